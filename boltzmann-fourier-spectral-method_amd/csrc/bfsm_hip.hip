// bfsm_hip.hip -- gfx950 kernels (instantiations of the bodies in bfsm_core.hpp) and the C-ABI of include/bfsm.h.
// Built with: hipcc --offload-arch=gfx950 -O3 -fPIC -shared  (see ../Makefile).  No rocFFT / hipFFT / MFMA.
#include <hip/hip_runtime.h>

#include <atomic>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "bfsm_pipeline.hpp"
#include "bfsm_generic.hpp"

#ifndef BFSM_F32_N64_WAVES
#define BFSM_F32_N64_WAVES 4
#endif

// Builds that compute WRONG results on purpose (knock-outs: no barriers / no LDS traffic / no stores / no butterflies) or
// carry instrumentation exist for timing experiments only (tools/).  They must be asked for explicitly: a stray -D or a
// macro-name collision in an embedding build cannot silently produce a library without barriers.  A tools build says so in
// bfsm_backend_name().
#if (defined(BFSM_KO_SYNC) || defined(BFSM_KO_LDS) || defined(BFSM_KO_STORE) || defined(BFSM_KO_DFT) || defined(BFSM_KA_BARRIER_TIMES)) && \
    !defined(BFSM_TOOLS_BUILD)
#error "BFSM_KO_* / BFSM_KA_BARRIER_TIMES are for tools-only builds: add -DBFSM_TOOLS_BUILD (the library then reports itself as a tools build)"
#endif

namespace bfsm {

// ---- device execution context ---------------------------------------------------------------------------------
struct DevCtx {
    unsigned char* smem;
    __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
    __device__ __forceinline__ int bx() const { return (int)blockIdx.x; }
    __device__ __forceinline__ int by() const { return (int)blockIdx.y; }
    __device__ __forceinline__ int bz() const { return (int)blockIdx.z; }
    __device__ __forceinline__ int gx() const { return (int)gridDim.x; }
    __device__ __forceinline__ int gy() const { return (int)gridDim.y; }
    __device__ __forceinline__ void sync() const {
#ifdef BFSM_KO_SYNC     // knock-out builds (tools only): timing experiments, wrong results
        return;
#endif
        __syncthreads();
    }
    // Waits until every vector-memory load issued so far has returned.  Placed in front of a loop that streams stores:
    // the compiler's wait-count pass merges the loop pre-header's pending loads (the f_hat plane, needed by the first
    // iteration) into the loop header's state and would otherwise emit s_waitcnt vmcnt(0) at the TOP OF EVERY ITERATION --
    // on gfx9 that counter also covers stores, so every wave would drain its own previous store burst before it starts
    // the next direction.  With the pre-header drained the header only waits for what the back edge really carries.
    __device__ __forceinline__ void drain_loads() const { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0), other counters free
    // nothing is scheduled across this point (bounds how many loads the compiler keeps in flight, i.e. registers)
    __device__ __forceinline__ void sched_fence() const { __builtin_amdgcn_sched_barrier(0); }
#ifdef BFSM_KA_BARRIER_TIMES     // instrumented build (tools only): cycles every wave spends inside each barrier of KA's pair loop
    __device__ __forceinline__ unsigned long long clk() const { return __builtin_readcyclecounter(); }
    __device__ __forceinline__ void dbg_add(int i, unsigned long long v) const;
#endif
    // keeps a loaded value (and therefore its load) alive up to this point without using it
    template <class T>
    __device__ __forceinline__ void keep_alive(T v) const { asm volatile("" ::"v"(v)); }
    // a wave-uniform value the optimiser may not reason about: loads addressed through it are neither hoisted out of
    // loops nor merged, so they occupy SGPRs only where they are used
    __device__ __forceinline__ int opaque(int v) const {
        asm volatile("" : "+s"(v));
        return v;
    }
    // the same for a per-lane integer
    __device__ __forceinline__ int opaque_v(int v) const {
        asm volatile("" : "+v"(v));
        return v;
    }
    // Lane byte offset of the row accessors below, re-materialised where it is called (inside a loop): hoisted out of
    // the loop as a zero-extended 64-bit value it defeats the saddr + voffset addressing form (one 64-bit VALU add and a
    // VGPR pair per access instead)
    __device__ __forceinline__ unsigned lane_off(unsigned v) const {
        asm volatile("" : "+v"(v));
        return v;
    }
    // a per-lane value the optimiser may not trace back to its origin (no common-subexpression sharing through it)
    template <class T>
    __device__ __forceinline__ cx<T> opaque_cx(cx<T> v) const {
        asm volatile("" : "+v"(v.x), "+v"(v.y));
        return v;
    }
    // value known to be equal across a wave when a row of n lanes covers whole waves: make it an SGPR so the
    // twiddle / phase-table loads that depend on it become scalar loads
    __device__ __forceinline__ int uniform(int v, int n) const {
        return (n % 64 == 0) ? __builtin_amdgcn_readfirstlane(v) : v;
    }
    template <class U>
    __device__ __forceinline__ U* lds() const { return reinterpret_cast<U*>(smem); }
    // Exchange-buffer accesses.  An 8-byte element (fp32 complex) is read as ONE ds_read_b64 per element: left to
    // itself the compiler pairs neighbouring reads into ds_read2_b64, which the LDS serves at half the rate of two
    // ds_read_b64 (8 cycles against 2 + 2 per wave-instruction, MI355X_MICROARCH.md LDS table).  A volatile access
    // is never paired; its order relative to the other exchange accesses is the program order anyway.
    template <class T>
    __device__ __forceinline__ cx<T> lds_ld(const cx<T>* p) const {
#ifdef BFSM_KO_LDS
        { cx<T> z = {(T)threadIdx.x, (T)1}; asm volatile("" : "+v"(z.x), "+v"(z.y)); return z; }
#endif
#ifndef BFSM_LDS_PAIRED
        if constexpr (sizeof(T) == 4) {
            typedef T vec2 __attribute__((ext_vector_type(2)));
            typedef const volatile vec2 __attribute__((address_space(3))) * lptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
            const vec2 r = *(lptr)(reinterpret_cast<const vec2*>(p));
#pragma clang diagnostic pop
            return cx<T>{r.x, r.y};
        }
#endif
        return *p;
    }
    template <class T>
    __device__ __forceinline__ void lds_st(cx<T>* p, cx<T> v) const {
#ifdef BFSM_KO_LDS      // knock-out builds (tools only): timing experiments, wrong results
        asm volatile("" ::"v"(v.x), "v"(v.y));
        return;
#endif
        *p = v;
    }
    // scalar element of the split (real / imaginary) exchange (fp64, N = 128): left to the compiler's pairing, which
    // measured faster than unpaired reads in that path
    template <class T>
    __device__ __forceinline__ T lds_ld_s(const T* p) const { return *p; }
    // Once-touched scratch (A1', A2', P'): nontemporal accesses keep the streams from evicting the small hot set
    // (f_hat planes, tables) out of L2.
    template <class T>
    __device__ __forceinline__ cx<T> ld_stream(const cx<T>* p) const {
        typedef T vec2 __attribute__((ext_vector_type(2)));
        const vec2 r = __builtin_nontemporal_load(reinterpret_cast<const vec2*>(p));
        return cx<T>{r.x, r.y};
    }
    template <class T>
    __device__ __forceinline__ void st_stream(cx<T>* p, cx<T> v) const {
        typedef T vec2 __attribute__((ext_vector_type(2)));
        vec2 r;
        r.x = v.x;
        r.y = v.y;
        __builtin_nontemporal_store(r, reinterpret_cast<vec2*>(p));
    }
    // The same with the address given as a row pointer plus a 32-bit per-lane byte offset.  UNI = the row pointer is
    // wave-uniform (a row of lanes covers whole waves): this shape
    // selects the scalar-base form of global_load / global_store (SGPR pair + one VGPR offset), so no 64-bit address
    // is built or kept in VGPRs per access.
    template <bool UNI, class T>
    __device__ __forceinline__ cx<T> ld_stream_at(const cx<T>* row, unsigned byte_off) const {
        if constexpr (!UNI) return ld_stream(reinterpret_cast<const cx<T>*>(reinterpret_cast<const unsigned char*>(row) + byte_off));
        typedef T vec2 __attribute__((ext_vector_type(2)));
        typedef const unsigned char __attribute__((address_space(1))) * gptr;
        typedef const vec2 __attribute__((address_space(1))) * gvec;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        gptr g = (gptr)(reinterpret_cast<const unsigned char*>(row));
        asm volatile("" : "+s"(g));          // keep the row pointer an SGPR pair of its own (no re-association)
#ifdef BFSM_PLAIN_LOADS     // A/B builds (tools only): ordinary loads for the streamed scratch
        const vec2 r = *(gvec)(g + byte_off);
#else
        const vec2 r = __builtin_nontemporal_load((gvec)(g + byte_off));
#endif
#pragma clang diagnostic pop
        return cx<T>{r.x, r.y};
    }
    template <bool UNI, class T>
    __device__ __forceinline__ void st_stream_at(cx<T>* row, unsigned byte_off, cx<T> v) const {
#ifdef BFSM_KO_STORE
        asm volatile("" ::"v"(v.x), "v"(v.y));
        return;
#endif
        if constexpr (!UNI) { st_stream(reinterpret_cast<cx<T>*>(reinterpret_cast<unsigned char*>(row) + byte_off), v); return; }
        typedef T vec2 __attribute__((ext_vector_type(2)));
        typedef unsigned char __attribute__((address_space(1))) * gptr;
        typedef vec2 __attribute__((address_space(1))) * gvec;
        vec2 r;
        r.x = v.x;
        r.y = v.y;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        gptr g = (gptr)(reinterpret_cast<unsigned char*>(row));
        asm volatile("" : "+s"(g));
#ifdef BFSM_PLAIN_STORES    // A/B builds (tools only): ordinary stores for the streamed scratch
        *(gvec)(g + byte_off) = r;
#else
        __builtin_nontemporal_store(r, (gvec)(g + byte_off));
#endif
#pragma clang diagnostic pop
    }
    // Interleaved scratch (ab_interleaved: element = {A1', A2'} of one grid point, 4 * sizeof(T) bytes): both values of a
    // point in ONE global_load / global_store of twice the width (dwordx4 in fp32), same scalar-base addressing form.
    template <bool UNI, class T>
    __device__ __forceinline__ void ld_stream_pair_at(const cx<T>* row, unsigned byte_off, cx<T>& v0, cx<T>& v1) const {
        typedef T vec4 __attribute__((ext_vector_type(4)));
        typedef const unsigned char __attribute__((address_space(1))) * gptr;
        typedef const vec4 __attribute__((address_space(1))) * gvec;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        gptr g = (gptr)(reinterpret_cast<const unsigned char*>(row));
        if constexpr (UNI) asm volatile("" : "+s"(g));
        const vec4 r = __builtin_nontemporal_load((gvec)(g + byte_off));
#pragma clang diagnostic pop
        v0 = cx<T>{r.x, r.y};
        v1 = cx<T>{r.z, r.w};
    }
    template <bool UNI, class T>
    __device__ __forceinline__ void st_stream_pair_at(cx<T>* row, unsigned byte_off, cx<T> v0, cx<T> v1) const {
#ifdef BFSM_KO_STORE
        asm volatile("" ::"v"(v0.x), "v"(v0.y), "v"(v1.x), "v"(v1.y));
        return;
#endif
        typedef T vec4 __attribute__((ext_vector_type(4)));
        typedef unsigned char __attribute__((address_space(1))) * gptr;
        typedef vec4 __attribute__((address_space(1))) * gvec;
        vec4 r;
        r.x = v0.x;
        r.y = v0.y;
        r.z = v1.x;
        r.w = v1.y;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        gptr g = (gptr)(reinterpret_cast<unsigned char*>(row));
        if constexpr (UNI) asm volatile("" : "+s"(g));
        __builtin_nontemporal_store(r, (gvec)(g + byte_off));
#pragma clang diagnostic pop
    }
    // Cross-lane exchange of a distributed line transform whose 8 threads sit at lane bits 3..5 of ONE wave (A/B build
    // BFSM_KA_XLANE): in v[k1] (k1 < 16) of thread u = lane >> 3; out v[q * 8 + uu] = (thread uu's v[u + 8 q]) -- two 8 x 8
    // register <-> lane transposes, one exchange step per lane bit: v_permlane32_swap (bit 5), v_permlane16_swap (bit 4),
    // DPP row_ror:8 + select (bit 3).  No LDS, no barrier.  The sequence validated in tools/micro/xlane_exchange.hip (inline
    // assembly: the swap builtins were miscompiled in this use; s_nop 1 = the wait states behind a VALU write of an operand).
    template <int BIT>
    __device__ __forceinline__ void xstep(float& a, float& b, bool bit_set) const {
        if constexpr (BIT == 32) {
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        } else if constexpr (BIT == 16) {
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        } else {
            float pa, pb;
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_ror:8 row_mask:0xf bank_mask:0xf"
                         : "=&v"(pa), "=&v"(pb) : "v"(a), "v"(b));
            const float na = bit_set ? pb : a, nb = bit_set ? b : pa;
            a = na;
            b = nb;
        }
    }
    __device__ __forceinline__ void xlane_transpose8(cx<float>* v) const {
        const int lane = (int)threadIdx.x & 63;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            cx<float>* r = v + 8 * q;          // block k1 = 8 q + j: transpose register index j <-> thread index u
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(j & 4)) { xstep<32>(r[j].x, r[j | 4].x, lane & 32); xstep<32>(r[j].y, r[j | 4].y, lane & 32); }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(j & 2)) { xstep<16>(r[j].x, r[j | 2].x, lane & 16); xstep<16>(r[j].y, r[j | 2].y, lane & 16); }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (!(j & 1)) { xstep<8>(r[j].x, r[j | 1].x, lane & 8); xstep<8>(r[j].y, r[j | 1].y, lane & 8); }
        }
    }
    __device__ __forceinline__ void xlane_transpose8(cx<double>*) const {}    // (no double-precision geometry uses it)
    // cacheable (plain) variants of the row + lane-offset accessors
    template <bool UNI, class T>
    __device__ __forceinline__ cx<T> ld_at(const cx<T>* row, unsigned byte_off) const {
        if constexpr (!UNI) return *reinterpret_cast<const cx<T>*>(reinterpret_cast<const unsigned char*>(row) + byte_off);
        typedef T vec2 __attribute__((ext_vector_type(2)));
        typedef const unsigned char __attribute__((address_space(1))) * gptr;
        typedef const vec2 __attribute__((address_space(1))) * gvec;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        gptr g = (gptr)(reinterpret_cast<const unsigned char*>(row));
        asm volatile("" : "+s"(g));
        const vec2 r = *(gvec)(g + byte_off);
#pragma clang diagnostic pop
        return cx<T>{r.x, r.y};
    }
    template <bool UNI, class T>
    __device__ __forceinline__ void st_at(cx<T>* row, unsigned byte_off, cx<T> v) const {
        if constexpr (!UNI) { *reinterpret_cast<cx<T>*>(reinterpret_cast<unsigned char*>(row) + byte_off) = v; return; }
        typedef T vec2 __attribute__((ext_vector_type(2)));
        typedef unsigned char __attribute__((address_space(1))) * gptr;
        typedef vec2 __attribute__((address_space(1))) * gvec;
        vec2 r;
        r.x = v.x;
        r.y = v.y;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        gptr g = (gptr)(reinterpret_cast<unsigned char*>(row));
        asm volatile("" : "+s"(g));
        *(gvec)(g + byte_off) = r;
#pragma clang diagnostic pop
    }
    // one real of a row (cacheable): wave-uniform row pointer + 32-bit lane byte offset (the L2 warm-up touches)
    template <bool UNI, class T>
    __device__ __forceinline__ T ld_real_at(const cx<T>* row, unsigned byte_off) const {
        typedef const unsigned char __attribute__((address_space(1))) * gptr;
        typedef const T __attribute__((address_space(1))) * gval;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        gptr g = (gptr)(reinterpret_cast<const unsigned char*>(row));
        if constexpr (UNI) asm volatile("" : "+s"(g));
        const T r = *(gval)(g + byte_off);
#pragma clang diagnostic pop
        return r;
    }
    // read-only table element through the constant address space: with a wave-uniform address the compiler
    // emits s_load (scalar data cache, SGPR result) even when the kernel also stores to global memory
    template <class T>
    __device__ __forceinline__ cx<T> ldc(const cx<T>* p) const {
        typedef const T __attribute__((address_space(4))) * cptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
        cptr q = (cptr)(reinterpret_cast<const T*>(p));
#pragma clang diagnostic pop
        return cx<T>{q[0], q[1]};
    }
};

#ifdef BFSM_KA_BARRIER_TIMES
__device__ unsigned long long bfsm_dbg_counters[32];
__device__ __forceinline__ void DevCtx::dbg_add(int i, unsigned long long v) const {
    if ((threadIdx.x & 63) == 0) atomicAdd(&bfsm_dbg_counters[i], v);
}
extern "C" int bfsm_debug_counters(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(bfsm_dbg_counters), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(bfsm_dbg_counters), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

template <K kind, int N>
constexpr int kernel_threads() {
    if (kind == K::GainInv && pair_tile<N>()) return pair_threads<N>();      // two tiles side by side (N = 32)
    return kind == K::Reduce ? 256 : (is_line_kind(kind) ? Wg<N>::LINE_THREADS : Wg<N>::THREADS);
}
template <K kind, int N, typename T>
constexpr size_t kernel_lds_bytes() {
    if (kind == K::GainInv && pair_tile<N>()) return pair_lds_bytes<N, T>();
    if (kind == K::GainInv && ka_xlane<N, T>()) return ka_xlane_lds_bytes<N, T>();
    if (kind == K::GainFwd) return kc_lds_bytes<N, T>();
    if (kind == K::GainInvNyq) return gain_inv_lds_bytes<N, T>();
    return kind == K::Reduce ? 0 : (is_line_kind(kind) ? line_lds_bytes<N, T>() : tile_lds_bytes<N, T>());
}

// Minimum waves per SIMD the register allocator must leave room for.  N=64: a 512-thread workgroup is 2 waves per
// SIMD and its 65 KiB tile lets two workgroups share a CU's 160 KiB LDS, so ask for 4 (<= 128 VGPRs).
template <K kind, int N, typename T>
constexpr int kernel_min_waves() {
    if (kind == K::Reduce) return 1;
    if (N == 64) return sizeof(T) == 4 ? BFSM_F32_N64_WAVES : 4;   // fp32 tiles are 33 KiB: more workgroups fit
    if (N == 32) return 4;                                         // 128-thread workgroups: 8 per CU at <= 128 VGPRs
    if (N == 128 && is_line_kind(kind)) return sizeof(T) == 4 ? 4 : 2;   // fp64: 133 KiB of columns, one workgroup per CU
    return 1;
}

template <K kind, int N, typename T, class P>
__global__ void __launch_bounds__((kernel_threads<kind, N>()), (kernel_min_waves<kind, N, T>())) bfsm_kernel(const P prm) {
    extern __shared__ __align__(16) unsigned char bfsm_smem[];
    DevCtx ctx{bfsm_smem};
    if constexpr (kind == K::TileFwdReal) body_tile_fwd_real<N, T>(prm, ctx);
    else if constexpr (kind == K::LineFwd) body_line<N, -1, T>(prm, ctx);
    else if constexpr (kind == K::LineInv) body_line<N, +1, T>(prm, ctx);
    else if constexpr (kind == K::TileFwd) body_tile_c2c<N, -1, T>(prm, ctx);
    else if constexpr (kind == K::TileInv) body_tile_c2c<N, +1, T>(prm, ctx);
    else if constexpr (kind == K::GainInv && pair_tile<N>()) body_gain_inv_pair<N, T>(prm, ctx);
    else if constexpr (kind == K::GainInv) body_gain_inv<N, T>(prm, ctx);
    else if constexpr (kind == K::GainLine) body_gain_line<N, T>(prm, ctx);
    else if constexpr (kind == K::GainFwd) body_gain_fwd<N, T>(prm, ctx);
    else if constexpr (kind == K::Reduce) body_reduce<N, T>(prm, ctx);
    else if constexpr (kind == K::TailInv) body_tail_inv<N, T>(prm, ctx);
    else if constexpr (kind == K::TailLine) body_tail_line<N, T>(prm, ctx);
    else if constexpr (kind == K::GainLineAcc) body_gain_line_acc<N, T>(prm, ctx);
    else if constexpr (kind == K::NyqRows) body_nyq_rows<N, T>(prm, ctx);
    else if constexpr (kind == K::GainLineAccH) body_gain_line_acc_h<N, T>(prm, ctx);
    else if constexpr (kind == K::GainInvNyq) {          // only launched where nyq_rides_along<N>()
        if constexpr (nyq_rides_along<N>()) body_gain_inv_nyq<N, T>(prm, ctx);
    } else if constexpr (kind == K::GainInvTwo) {          // only launched on ab_interleaved geometries (Hermitian mode)
        if constexpr (ab_interleaved<N, T>()) body_gain_inv<N, T, false>(prm, ctx);
    }
}

// size-generic path (bfsm_generic.hpp): runtime sizes, 256 threads, dynamic LDS
template <GK kind, typename T, class P>
__global__ void __launch_bounds__(GEN_THREADS) bfsm_gen_kernel(const P prm) {
    extern __shared__ __align__(16) unsigned char bfsm_smem[];
    DevCtx ctx{bfsm_smem};
    if constexpr (kind == GK::Fft) body_gen_fft<T, false, GEN_C>(prm, ctx);
    else if constexpr (kind == GK::FftBig) body_gen_fft<T, true, GEN_C>(prm, ctx);
    else if constexpr (kind == GK::Plane) body_gen_plane<T>(prm, ctx);
    else if constexpr (kind == GK::Acc) body_gen_acc<T>(prm, ctx);
    else if constexpr (kind == GK::Combine) body_gen_combine<T>(prm, ctx);
    else if constexpr (kind == GK::Line3) body_gen_line3<T, GEN_C>(prm, ctx);
    else if constexpr (kind == GK::PlaneAcc) body_gen_plane_acc<T>(prm, ctx);
    else if constexpr (kind == GK::PlanePair) body_gen_plane_pair<T>(prm, ctx);
    else if constexpr (kind == GK::Fft8) body_gen_fft<T, false, 8>(prm, ctx);
    else if constexpr (kind == GK::FftBig8) body_gen_fft<T, true, 8>(prm, ctx);
    else if constexpr (kind == GK::Line38) body_gen_line3<T, 8>(prm, ctx);
}

// N = 16 whole-direction kernels: 256 threads, two padded cubes of LDS
template <SK kind, typename T, class P>
__global__ void __launch_bounds__(SMALL_THREADS) bfsm_small_kernel(const P prm) {
    extern __shared__ __align__(16) unsigned char bfsm_smem[];
    DevCtx ctx{bfsm_smem};
    if constexpr (kind == SK::Gain) body_small_gain<T>(prm, ctx);
    else if constexpr (kind == SK::Reduce) body_small_reduce<T>(prm, ctx);
}

// ---- HIP backend -----------------------------------------------------------------------------------------------
struct HipBackend {
    hipStream_t stream = nullptr;
    int device = 0;               // HIP ordinal this backend launches on (set by the handle)
    bool profile = false;
    hipError_t first_error = hipSuccess;
    const char* first_error_where = "";
    int first_error_line = 0;

    struct Rec { int kind; double bytes; hipEvent_t e0, e1; };
    std::vector<Rec> recs;        // launches of the current profiled evaluation
    std::vector<hipEvent_t> pool; // reusable events
    size_t pool_next = 0;
    int pend_kind = -1;
    double pend_bytes = 0;

    void note(hipError_t e, const char* where, int line) {
        if (e != hipSuccess && first_error == hipSuccess) { first_error = e; first_error_where = where; first_error_line = line; }
    }
#define BFSM_NOTE(call) note((call), #call, __LINE__)

    void* alloc(size_t bytes) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        note(e, "hipMalloc", __LINE__);
        if (e != hipSuccess) (void)hipGetLastError();   // reported through the status code: do not leave it sticky
        return e == hipSuccess ? p : nullptr;
    }
    void release(void* p) { BFSM_NOTE(hipFree(p)); }
    void upload(void* dst, const void* src, size_t bytes) { BFSM_NOTE(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); }

    hipEvent_t next_event() {
        if (pool_next == pool.size()) {
            hipEvent_t e;
            BFSM_NOTE(hipEventCreate(&e));
            pool.push_back(e);
        }
        return pool[pool_next++];
    }
    void begin_eval() { recs.clear(); pool_next = 0; }
    void mark(int kind, double bytes) { pend_kind = kind; pend_bytes = bytes; }

    // Common launcher: opts the kernel in to a large dynamic-LDS allocation (once per kernel instantiation and device),
    // brackets the launch with events when profiling, and takes the launch's OWN status from hipLaunchKernel: an earlier,
    // unrelated sticky error of the calling thread (a failed hipMalloc of another handle, the caller's own HIP calls) is
    // neither blamed on this launch nor consumed.
    void launch_any(const void* fn, int gx, int gy, int gz, int threads, size_t lds, const void* prm,
                    std::atomic<unsigned long long>& lds_opted_in, size_t lds_limit = 0) {
        if (gx <= 0 || gy <= 0 || gz <= 0) return;
        if (lds > 48 * 1024) {
            const unsigned long long bit = 1ull << (device & 63);
            if (!(lds_opted_in.load(std::memory_order_relaxed) & bit)) {
                BFSM_NOTE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_limit ? lds_limit : lds)));
                lds_opted_in.fetch_or(bit, std::memory_order_relaxed);
            }
        }
        Rec rec{pend_kind, pend_bytes, nullptr, nullptr};
        const bool timed = profile && pend_kind >= 0;
        if (timed) { rec.e0 = next_event(); rec.e1 = next_event(); BFSM_NOTE(hipEventRecord(rec.e0, stream)); }
        void* args[] = {const_cast<void*>(prm)};
        BFSM_NOTE(hipLaunchKernel(fn, dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3((unsigned)threads, 1, 1), args, lds, stream));
        if (timed) { BFSM_NOTE(hipEventRecord(rec.e1, stream)); recs.push_back(rec); }
        pend_kind = -1;
    }

    template <K kind, int N, typename T, class P>
    void launch_n(int gx, int gy, int gz, const P& prm) {
        static std::atomic<unsigned long long> opted{0};
        launch_any(reinterpret_cast<const void*>(bfsm_kernel<kind, N, T, P>), gx, gy, gz, kernel_threads<kind, N>(),
                   kernel_lds_bytes<kind, N, T>(), &prm, opted);
    }

    // N = 16 whole-direction kernels
    template <SK kind, typename T, class P>
    void launch_small(int gx, const P& prm) {
        static std::atomic<unsigned long long> opted{0};
        launch_any(reinterpret_cast<const void*>(bfsm_small_kernel<kind, T, P>), gx, 1, 1, SMALL_THREADS,
                   kind == SK::Reduce ? 256 * sizeof(double) : small_lds_bytes<T>(), &prm, opted);
    }

    // size-generic path; the LDS need depends on the transformed axis, so the opt-in asks for the whole CU's 160 KiB
    template <GK kind, typename T, class P>
    void launch_gen(int gx, int gy, int threads, size_t lds, const P& prm) {
        static std::atomic<unsigned long long> opted{0};
        launch_any(reinterpret_cast<const void*>(bfsm_gen_kernel<kind, T, P>), gx, gy, 1, threads, lds, &prm, opted,
                   (size_t)160 * 1024);
    }

    template <K kind, typename T, class P>
    void launch(int gx, int gy, int gz, const P& prm, int N) {
        if (gx <= 0 || gy <= 0 || gz <= 0) return;
        switch (N) {
            case 16: launch_n<kind, 16, T>(gx, gy, gz, prm); break;
            case 24: launch_n<kind, 24, T>(gx, gy, gz, prm); break;
            case 32: launch_n<kind, 32, T>(gx, gy, gz, prm); break;
            case 40: launch_n<kind, 40, T>(gx, gy, gz, prm); break;
            case 48: launch_n<kind, 48, T>(gx, gy, gz, prm); break;
            case 80: launch_n<kind, 80, T>(gx, gy, gz, prm); break;
            case 64: launch_n<kind, 64, T>(gx, gy, gz, prm); break;
            case 96: launch_n<kind, 96, T>(gx, gy, gz, prm); break;
            case 128: launch_n<kind, 128, T>(gx, gy, gz, prm); break;
            default: break;
        }
    }

    void destroy_events() {
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

}  // namespace bfsm

// ---- the handle ------------------------------------------------------------------------------------------------
struct bfsm_plan {
    bfsm_desc desc{};
    bfsm::HipBackend be;
    bfsm::Pipeline<double, bfsm::HipBackend>* p64 = nullptr;
    bfsm::Pipeline<float, bfsm::HipBackend>* p32 = nullptr;
    bfsm::GenericPipeline<double, bfsm::HipBackend>* g64 = nullptr;   // grids outside the fused pipeline's sizes
    bfsm::GenericPipeline<float, bfsm::HipBackend>* g32 = nullptr;
    bfsm::PlanInfo info;
    size_t G = 0;
    // calls fn(pipeline) on whichever of the four pipelines this handle owns
    template <class F>
    void with(F&& fn) {
        if (p64) fn(*p64); else if (p32) fn(*p32); else if (g64) fn(*g64); else if (g32) fn(*g32);
    }
    std::string err;
    bfsm_counters counters{};
    bool full_shard = true;
    // Work this handle has enqueued since its last bfsm_synchronize: one handle-owned event per distinct stream, recorded
    // behind the last call on that stream.  The stream value is kept as a KEY only (compared, never passed to HIP again),
    // so a stream the caller has destroyed in the meantime is harmless.
    struct Pending { hipStream_t key; hipEvent_t ev; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> spare_events;
};

static thread_local std::string g_create_error;

static int fail(bfsm_plan* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

// Makes the handle's device current for the duration of a call and puts the caller's device back afterwards: the
// library never changes the calling thread's current device as a side effect.
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;   // nothing to restore when it already is current
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// No C++ exception may cross the C boundary (bfsm.h: "functions never throw"): host-side allocation failures of the
// plan / table builders and anything unexpected become status codes.
#define BFSM_GUARDED(h, body)                                                                         \
    try { body }                                                                                      \
    catch (const std::bad_alloc&) { return fail((h), BFSM_ERR_NOMEM, "out of host memory"); }         \
    catch (const std::exception& ex) { return fail((h), BFSM_ERR_INVALID, std::string("internal error: ") + ex.what()); } \
    catch (...) { return fail((h), BFSM_ERR_INVALID, "internal error (unknown exception)"); }

static int check_hip(bfsm_plan* h, const char* where) {
    if (h->be.first_error == hipSuccess) return BFSM_OK;
    std::string m = std::string("HIP error: ") + hipGetErrorString(h->be.first_error) + " in " + h->be.first_error_where +
                    " at bfsm_hip.hip:" + std::to_string(h->be.first_error_line) + " (during " + where + ")";
    h->be.first_error = hipSuccess;
    return fail(h, BFSM_ERR_HIP, m);
}

extern "C" {

#ifdef BFSM_TOOLS_BUILD
const char* bfsm_backend_name(void) { return "HIP (tools build: timing experiments, results not valid)"; }
#else
const char* bfsm_backend_name(void) { return "HIP"; }
#endif
int bfsm_version(void) { return BFSM_VERSION; }

const char* bfsm_last_error(bfsm_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int create_impl(const bfsm_desc* desc, bfsm_handle* out, bfsm_plan*& h) {
    std::string err;
    int rc = bfsm::validate_desc(*desc, err);
    if (rc != BFSM_OK) return fail(nullptr, rc, err);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, BFSM_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (desc->device < 0 || desc->device >= ndev) return fail(nullptr, BFSM_ERR_INVALID, "device ordinal out of range");
    DeviceGuard guard(desc->device);
    if (guard.err != hipSuccess) return fail(nullptr, BFSM_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.err));
    h = new bfsm_plan();
    h->desc = *desc;
    h->be.profile = (desc->flags & BFSM_FLAG_PROFILE) != 0;
    h->be.device = desc->device;
    h->G = (size_t)desc->nvx * desc->nvy * desc->nvz;
    if (!bfsm::fused_grid(*desc)) {
        if (desc->precision == BFSM_F64) {
            h->g64 = new bfsm::GenericPipeline<double, bfsm::HipBackend>();
            rc = h->g64->init(*desc, &h->be, err);
            h->info = h->g64->plan;
        } else {
            h->g32 = new bfsm::GenericPipeline<float, bfsm::HipBackend>();
            rc = h->g32->init(*desc, &h->be, err);
            h->info = h->g32->plan;
        }
    } else if (desc->precision == BFSM_F64) {
        h->p64 = new bfsm::Pipeline<double, bfsm::HipBackend>();
        rc = h->p64->init(*desc, &h->be, err);
        h->info = h->p64->plan;
    } else {
        h->p32 = new bfsm::Pipeline<float, bfsm::HipBackend>();
        rc = h->p32->init(*desc, &h->be, err);
        h->info = h->p32->plan;
    }
    // the descriptor's host arrays are not referenced after create
    h->desc.gl_nodes = h->desc.gl_wts = h->desc.sph_wts = h->desc.sx = h->desc.sy = h->desc.sz = nullptr;
    h->full_shard = h->info.full_begin == 0 && h->info.full_end == (long long)desc->n_gl * desc->n_sph;
    if (rc == BFSM_OK) rc = check_hip(h, "bfsm_create");
    else if (h->be.first_error != hipSuccess) { (void)check_hip(h, "bfsm_create"); err = h->err; }
    if (rc != BFSM_OK) {
        g_create_error = h->err.empty() ? err : h->err;
        return rc;
    }
    *out = h;
    h = nullptr;   // ownership passed to the caller
    return BFSM_OK;
}

int bfsm_create(const bfsm_desc* desc, bfsm_handle* out) {
    if (!desc || !out) return fail(nullptr, BFSM_ERR_INVALID, "null argument");
    *out = nullptr;
    bfsm_plan* h = nullptr;   // whatever create_impl leaves here (error or exception) is torn down
    int rc;
    try {
        rc = create_impl(desc, out, h);
    } catch (const std::bad_alloc&) {
        rc = fail(nullptr, BFSM_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& ex) {
        rc = fail(nullptr, BFSM_ERR_INVALID, std::string("internal error: ") + ex.what());
    } catch (...) {
        rc = fail(nullptr, BFSM_ERR_INVALID, "internal error (unknown exception)");
    }
    if (h) {
        const std::string keep = g_create_error;   // bfsm_destroy must not disturb the message
        bfsm_destroy(h);
        g_create_error = keep;
    }
    return rc;
}

// Common prologue of the entry points (after the DeviceGuard): the stream of this call.
static int enter(bfsm_plan* h, const DeviceGuard& g, void* stream) {
    if (g.err != hipSuccess) return fail(h, BFSM_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(g.err));
    h->be.stream = (hipStream_t)stream;
    return BFSM_OK;
}

// Common epilogue of the entry points that enqueue work: the launch status of this call, then one event record behind
// the call so that bfsm_synchronize can wait for it without ever touching the caller's stream handle again.  Never
// blocks: one capture-status query (hipStreamIsCapturing, host-side) and one hipEventRecord (hipEventCreate on the first
// use of a stream); nothing is recorded while the stream is being captured into a graph (a captured launch does not run,
// and a captured event could not be waited for) -- replays of such a graph are the caller's to synchronise.
static int leave(bfsm_plan* h, const char* where) {
    int rc = check_hip(h, where);
    if (rc) return rc;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->be.stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) return BFSM_OK;
    bfsm_plan::Pending* slot = nullptr;
    for (auto& pe : h->pending) if (pe.key == h->be.stream) slot = &pe;
    if (!slot) {          // one entry per distinct stream since the last bfsm_synchronize: no work is ever left untracked
        hipEvent_t ev = nullptr;
        if (!h->spare_events.empty()) { ev = h->spare_events.back(); h->spare_events.pop_back(); }
        else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            return fail(h, BFSM_ERR_HIP, std::string("hipEventCreateWithFlags failed (during ") + where + ")");
        h->pending.push_back({h->be.stream, ev});
        slot = &h->pending.back();
    }
    hipError_t e = hipEventRecord(slot->ev, h->be.stream);
    if (e != hipSuccess) return fail(h, BFSM_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(e) + " (during " + where + ")");
    return BFSM_OK;
}

int bfsm_gain_partial(bfsm_handle h, const double* f_dev, void* stream) {
    if (!h) return BFSM_ERR_INVALID;
    BFSM_GUARDED(h,
        DeviceGuard g(h->desc.device);
        int rc = enter(h, g, stream);
        if (rc) return rc;
        if (!f_dev) return fail(h, BFSM_ERR_INVALID, "null f");
        h->be.begin_eval();
        h->with([&](auto& p) { p.gain_partial(f_dev); });
        return leave(h, "bfsm_gain_partial");
    )
}

int bfsm_finish(bfsm_handle h, double* Q_dev, const double* f_dev, void* stream) {
    if (!h) return BFSM_ERR_INVALID;
    BFSM_GUARDED(h,
        DeviceGuard g(h->desc.device);
        int rc = enter(h, g, stream);
        if (rc) return rc;
        if (!f_dev || !Q_dev) return fail(h, BFSM_ERR_INVALID, "null f or Q");
        h->with([&](auto& p) { p.finish(Q_dev, f_dev); });
        return leave(h, "bfsm_finish");
    )
}

int bfsm_finish_partial(bfsm_handle h, double* Q_dev, const double* f_dev, int with_loss, void* stream) {
    if (!h) return BFSM_ERR_INVALID;
    BFSM_GUARDED(h,
        DeviceGuard g(h->desc.device);
        int rc = enter(h, g, stream);
        if (rc) return rc;
        if (!f_dev || !Q_dev) return fail(h, BFSM_ERR_INVALID, "null f or Q");
        h->with([&](auto& p) { p.finish(Q_dev, f_dev, with_loss != 0); });
        return leave(h, "bfsm_finish_partial");
    )
}

int bfsm_collide_batch_partial_async(bfsm_handle h, double* Q_dev, const double* f_dev, int n_batch, int with_loss, void* stream) {
    if (!h) return BFSM_ERR_INVALID;
    BFSM_GUARDED(h,
        DeviceGuard g(h->desc.device);
        int rc = enter(h, g, stream);
        if (rc) return rc;
        if (!f_dev || !Q_dev) return fail(h, BFSM_ERR_INVALID, "null f or Q");
        int cap = 1;
        h->with([&](auto& p) { cap = p.max_batch; });
        if (n_batch < 1 || n_batch > cap)
            return fail(h, BFSM_ERR_INVALID, "n_batch must be in [1, max_batch of the descriptor]");
        h->be.begin_eval();
        if (h->g64 || h->g32) {      // size-generic path: all members through every launch of the fused sequence, else one by one
            bool together = false;
            h->with([&](auto& p) { together = p.batch_together(); });
            if (together) h->with([&](auto& p) { p.gain_partial(f_dev, n_batch); p.finish(Q_dev, f_dev, with_loss != 0, n_batch); });
            else
                for (int i = 0; i < n_batch; ++i)
                    h->with([&](auto& p) { p.gain_partial(f_dev + (size_t)i * h->G); p.finish(Q_dev + (size_t)i * h->G, f_dev + (size_t)i * h->G, with_loss != 0); });
        } else {
            h->with([&](auto& p) {
                // a batch of one on a single-evaluation N = 16 handle takes the same whole-direction kernels as bfsm_collide
                if (p.small_path(n_batch)) { p.collide_small(Q_dev, f_dev, with_loss != 0); return; }
                const bool fu = p.fuse_reduce(); p.gain_partial(f_dev, n_batch, !fu); p.finish(Q_dev, f_dev, with_loss != 0, n_batch, fu);
            });
        }
        return leave(h, "bfsm_collide_batch");
    )
}

int bfsm_collide_batch_async(bfsm_handle h, double* Q_dev, const double* f_dev, int n_batch, void* stream) {
    if (!h) return BFSM_ERR_INVALID;
    if (!h->full_shard) return fail(h, BFSM_ERR_INVALID, "batched evaluation needs a handle that owns all directions; use bfsm_collide_batch_partial_async + a sum over the ranks");
    return bfsm_collide_batch_partial_async(h, Q_dev, f_dev, n_batch, 1, stream);
}

int bfsm_collide_batch(bfsm_handle h, double* Q_dev, const double* f_dev, int n_batch) {
    int rc = bfsm_collide_batch_async(h, Q_dev, f_dev, n_batch, nullptr);
    if (rc) return rc;
    return bfsm_synchronize(h);
}

int bfsm_collide_async(bfsm_handle h, double* Q_dev, const double* f_dev, void* stream) {
    if (!h) return BFSM_ERR_INVALID;
    if (!h->full_shard)
        return fail(h, BFSM_ERR_INVALID, "bfsm_collide needs a handle that owns all directions; use gain_partial + reduce + finish");
    return bfsm_collide_partial_async(h, Q_dev, f_dev, 1, stream);
}

int bfsm_collide_partial_async(bfsm_handle h, double* Q_dev, const double* f_dev, int with_loss, void* stream) {
    if (!h) return BFSM_ERR_INVALID;
    BFSM_GUARDED(h,
        DeviceGuard g(h->desc.device);
        int rc = enter(h, g, stream);
        if (rc) return rc;
        if (!f_dev || !Q_dev) return fail(h, BFSM_ERR_INVALID, "null f or Q");
        h->be.begin_eval();
        // gain kernels, then the tail; with few slabs the reduce is fused into its first kernel (qhat is not written then)
        h->with([&](auto& p) {
            if (p.small_path(1)) { p.collide_small(Q_dev, f_dev, with_loss != 0); return; }   // N = 16: whole-direction kernels
            const bool fu = p.fuse_reduce();
            p.gain_partial(f_dev, 1, !fu);
            p.finish(Q_dev, f_dev, with_loss != 0, 1, fu);
        });
        return leave(h, "bfsm_collide_partial");
    )
}

int bfsm_collide(bfsm_handle h, double* Q_dev, const double* f_dev) {
    int rc = bfsm_collide_async(h, Q_dev, f_dev, nullptr);
    if (rc) return rc;
    return bfsm_synchronize(h);
}

int bfsm_synchronize(bfsm_handle h) {
    if (!h) return BFSM_ERR_INVALID;
    BFSM_GUARDED(h,
        DeviceGuard g(h->desc.device);
        hipError_t e = g.err;
        // every stream a call on this handle was given, not only the last one; the list is cleared whatever happens, so
        // one failure does not poison the later calls
        for (auto& pe : h->pending) {
            const hipError_t ei = (g.err == hipSuccess) ? hipEventSynchronize(pe.ev) : g.err;
            if (e == hipSuccess) e = ei;
            h->spare_events.push_back(pe.ev);
        }
        h->pending.clear();
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(h, BFSM_ERR_HIP, std::string("bfsm_synchronize: ") + hipGetErrorString(e)); }
        return BFSM_OK;
    )
}

void* bfsm_qhat_buffer(bfsm_handle h, size_t* n_elems, int* precision) {
    if (!h) return nullptr;
    if (n_elems) *n_elems = 2 * h->G;
    if (precision) *precision = h->info.precision;
    void* q = nullptr;
    h->with([&](auto& p) { q = (void*)p.qhat; });
    return q;
}

int bfsm_fft3d(bfsm_handle h, void* data_dev, int batch, int sign) {
    if (!h) return BFSM_ERR_INVALID;
    BFSM_GUARDED(h,
        DeviceGuard g(h->desc.device);
        int rc = enter(h, g, nullptr);
        if (rc) return rc;
        if (!data_dev || batch < 1 || batch > 65535 || (sign != 1 && sign != -1))
            return fail(h, BFSM_ERR_INVALID, "bad fft3d argument (null data, batch outside [1, 65535] or sign not +-1)");
        if (h->p64) h->p64->fft3d((bfsm::cx<double>*)data_dev, batch, sign);
        else if (h->p32) h->p32->fft3d((bfsm::cx<float>*)data_dev, batch, sign);
        else if (h->g64) h->g64->fft3d((bfsm::cx<double>*)data_dev, batch, sign);
        else h->g32->fft3d((bfsm::cx<float>*)data_dev, batch, sign);
        rc = leave(h, "bfsm_fft3d");
        if (rc) return rc;
        return bfsm_synchronize(h);
    )
}

int bfsm_get_counters(bfsm_handle h, bfsm_counters* out) {
    if (!h || !out) return BFSM_ERR_INVALID;
    bfsm_counters c{};
    c.alg_bytes_per_eval = bfsm::alg_bytes_per_eval(h->info);
    c.n_chunks = (int)h->info.chunks.size();
    c.chunk_dirs = h->info.largest_chunk;
    c.n_dirs = h->info.n_dirs();
    c.moved_bytes_per_eval = bfsm::moved_bytes_per_eval(h->info);
    c.exact_reductions = h->info.exact_reductions ? 1 : 0;
    c.antipodal_merged = h->info.antipodal ? 1 : 0;
    if (h->be.profile && !h->be.recs.empty()) {
        DeviceGuard g(h->desc.device);
        hipError_t e = g.err;
        if (e == hipSuccess) e = hipEventSynchronize(h->be.recs.back().e1);   // the handle's own event, not the caller's stream
        if (e != hipSuccess) return fail(h, BFSM_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
        for (const auto& r : h->be.recs) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, r.e0, r.e1);
            if (e != hipSuccess) return fail(h, BFSM_ERR_HIP, std::string("hipEventElapsedTime: ") + hipGetErrorString(e));
            if (r.kind >= 0 && r.kind < BFSM_K_COUNT) {
                c.kernel_ms[r.kind] += ms;
                c.kernel_alg_bytes[r.kind] += r.bytes;
                c.kernel_launches[r.kind] += 1;
            }
        }
    }
    *out = c;
    return BFSM_OK;
}

int bfsm_destroy(bfsm_handle h) {
    if (!h) return BFSM_OK;
    DeviceGuard g(h->desc.device);
    (void)hipDeviceSynchronize();
    if (h->p64) { h->p64->destroy(); delete h->p64; }
    if (h->p32) { h->p32->destroy(); delete h->p32; }
    if (h->g64) { h->g64->destroy(); delete h->g64; }
    if (h->g32) { h->g32->destroy(); delete h->g32; }
    h->be.destroy_events();
    for (auto& pe : h->pending) (void)hipEventDestroy(pe.ev);
    for (hipEvent_t ev : h->spare_events) (void)hipEventDestroy(ev);
    delete h;
    return BFSM_OK;
}

}  // extern "C"
