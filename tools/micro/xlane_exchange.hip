// Cross-lane ("wavefront shuffle") exchange against the LDS exchange, in isolation, on the N = 128 fp32 geometry of KA
// (16 points per thread, 8 threads per line, 1024-thread workgroups, one per CU, 129 KiB of LDS reserved in every mode so
// that the occupancy is the same).  One iteration = one distributed 128-point line transform with the library's own
// register butterflies (csrc/bfsm_core.hpp): radix-16 -> exchange among the 8 threads of the line -> 2 x radix-8 with the
// inter-step twiddles folded in.
//   mode 0  LDS exchange, the product's form: partners of a line sit in 8 different waves (tid = u*128 + p);
//           16 ds_write_b64 + barrier + 16 ds_read_b64 + barrier
//   mode 1  cross-lane exchange: the 8 partners sit in one wave at lane bits 3..5 (lane = p_lo + 8 u); the 8 x 8 transposes
//           are done in registers with v_permlane32_swap (lane bit 5), v_permlane16_swap (bit 4) and DPP row_ror:8 +
//           v_cndmask (bit 3); no LDS traffic, no barrier
//   mode 2  butterflies only (no exchange: wrong transform, the arithmetic floor)
//   mode 3  mode 0 without the butterflies          mode 4  mode 1 without the butterflies
//   mode 5  LDS exchange through two planes of floats with ds_write_addtid_b32 / ds_read_addtid_b32 (M0-relative, no
//           address VGPR: 2 cycles per stored dword instead of 6 per ds_write_b64; reads 2 x b32 instead of 1 x b64)
//   mode 6  mode 5 without the butterflies
// Both exchanges are checked against a host DFT before timing.
// build: hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -I../../boltzmann-fourier-spectral-method_amd/csrc -o xlane_exchange xlane_exchange.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bfsm_core.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

using bfsm::cx;
constexpr int N = 128, E = 16, TT = 8, LS = N + 1;

struct Ctx {   // the two accessors the twiddle holder needs
    template <class T> __device__ __forceinline__ cx<T> ldc(const cx<T>* p) const { return *p; }
    __device__ __forceinline__ int opaque(int v) const { return v; }
    __device__ __forceinline__ int opaque_v(int v) const { return v; }
};

// exchange between the two lanes that differ in lane bit `BIT` (8, 16 or 32): lanes with the bit clear give b and take
// the partner's a; lanes with the bit set give a and take the partner's b  (one step of a register <-> lane transpose).
// Written as inline assembly: the instruction sequence is exactly what is timed (s_nop 1 = the two wait states the
// permlane swaps need behind a VALU write of an operand).
template <int BIT>
__device__ __forceinline__ void xstep(float& a, float& b, bool bit_set) {
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

// in: v[k1] of thread u;  out: w2[q*8 + uu] = v_of_thread_uu[u + 8 q]   (the exchange of fft_line_np), u = lane bits 3..5
__device__ __forceinline__ void xlane_exchange(cx<float>* v, int lane) {
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

// split-plane exchange (modes 5, 6): real and imaginary parts in two planes of floats, stored and loaded with the
// M0-relative ds_*_addtid_b32 forms (address = M0 + offset + 4 * lane: no address VGPR; a store then costs 2 cycles per
// dword instead of 6 per ds_write_b64)
__device__ __forceinline__ void addtid_store16(unsigned m0, float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float v8, float v9, float v10, float v11, float v12, float v13, float v14, float v15) {
    asm volatile("s_mov_b32 m0, %16\n\ts_nop 0\n\t"
        "ds_write_addtid_b32 %0 offset:0\n\t"
        "ds_write_addtid_b32 %1 offset:4128\n\t"
        "ds_write_addtid_b32 %2 offset:8256\n\t"
        "ds_write_addtid_b32 %3 offset:12384\n\t"
        "ds_write_addtid_b32 %4 offset:16512\n\t"
        "ds_write_addtid_b32 %5 offset:20640\n\t"
        "ds_write_addtid_b32 %6 offset:24768\n\t"
        "ds_write_addtid_b32 %7 offset:28896\n\t"
        "ds_write_addtid_b32 %8 offset:33024\n\t"
        "ds_write_addtid_b32 %9 offset:37152\n\t"
        "ds_write_addtid_b32 %10 offset:41280\n\t"
        "ds_write_addtid_b32 %11 offset:45408\n\t"
        "ds_write_addtid_b32 %12 offset:49536\n\t"
        "ds_write_addtid_b32 %13 offset:53664\n\t"
        "ds_write_addtid_b32 %14 offset:57792\n\t"
        "ds_write_addtid_b32 %15 offset:61920\n\t"
        :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "v"(v8), "v"(v9), "v"(v10), "v"(v11), "v"(v12), "v"(v13), "v"(v14), "v"(v15), "s"(m0) : "memory");
}
__device__ __forceinline__ void addtid_load16(unsigned m0, float* r) {
    asm volatile("s_mov_b32 m0, %16\n\ts_nop 0\n\t"
        "ds_read_addtid_b32 %0 offset:0\n\t"
        "ds_read_addtid_b32 %1 offset:516\n\t"
        "ds_read_addtid_b32 %2 offset:1032\n\t"
        "ds_read_addtid_b32 %3 offset:1548\n\t"
        "ds_read_addtid_b32 %4 offset:2064\n\t"
        "ds_read_addtid_b32 %5 offset:2580\n\t"
        "ds_read_addtid_b32 %6 offset:3096\n\t"
        "ds_read_addtid_b32 %7 offset:3612\n\t"
        "ds_read_addtid_b32 %8 offset:33024\n\t"
        "ds_read_addtid_b32 %9 offset:33540\n\t"
        "ds_read_addtid_b32 %10 offset:34056\n\t"
        "ds_read_addtid_b32 %11 offset:34572\n\t"
        "ds_read_addtid_b32 %12 offset:35088\n\t"
        "ds_read_addtid_b32 %13 offset:35604\n\t"
        "ds_read_addtid_b32 %14 offset:36120\n\t"
        "ds_read_addtid_b32 %15 offset:36636\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]) : "s"(m0) : "memory");
}
template <int MODE>
__global__ void __launch_bounds__(1024, 1) probe(const cx<float>* tw, const cx<float>* in, cx<float>* out, int iters) {
    extern __shared__ __align__(16) unsigned char smem[];
    cx<float>* lds = reinterpret_cast<cx<float>*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr bool XL = (MODE == 1 || MODE == 4), DFT = (MODE <= 2 || MODE == 5);
    // line id and position inside the line
    const int u = XL ? (lane >> 3) & 7 : __builtin_amdgcn_readfirstlane(tid / N);
    const int p = XL ? (tid >> 6) * 8 + (lane & 7) : tid % N;          // column = line id inside the workgroup
    Ctx ctx;
    bfsm::Twiddles<N, float> twr;
    twr.load(tw, u, ctx);
    cx<float> v[E];
    const size_t base = ((size_t)blockIdx.x * N + p) * N;               // line-major input: x[line][n]
#pragma unroll
    for (int m = 0; m < E; ++m) v[m] = in[base + u + TT * m];
    for (int it = 0; it < iters; ++it) {
        if (DFT) bfsm::SmallDft<E, -1, float>::run(v);
        cx<float> w2[E];
        if constexpr (MODE == 0 || MODE == 3) {
            __syncthreads();
#pragma unroll
            for (int k1 = 0; k1 < E; ++k1) lds[(k1 * TT + u) * LS + p] = v[k1];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int uu = 0; uu < TT; ++uu) w2[q * TT + uu] = lds[((u + TT * q) * TT + uu) * LS + p];
        } else if constexpr (MODE == 5 || MODE == 6) {
            constexpr unsigned PLANE = N * LS * 4;
            const unsigned half = 64u * ((unsigned)(tid >> 6) & 1u);
            const unsigned wb = __builtin_amdgcn_readfirstlane(((unsigned)u * LS + half) * 4u);
            const unsigned rb = __builtin_amdgcn_readfirstlane(((unsigned)u * TT * LS + half) * 4u);
            __syncthreads();
            addtid_store16(wb, v[0].x, v[1].x, v[2].x, v[3].x, v[4].x, v[5].x, v[6].x, v[7].x, v[8].x, v[9].x, v[10].x, v[11].x, v[12].x, v[13].x, v[14].x, v[15].x);
            addtid_store16(wb + PLANE, v[0].y, v[1].y, v[2].y, v[3].y, v[4].y, v[5].y, v[6].y, v[7].y, v[8].y, v[9].y, v[10].y, v[11].y, v[12].y, v[13].y, v[14].y, v[15].y);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            float re[16], im[16];
            addtid_load16(rb, re);
            addtid_load16(rb + PLANE, im);
#pragma unroll
            for (int k = 0; k < E; ++k) w2[k] = {re[k], im[k]};
        } else if constexpr (XL) {
            xlane_exchange(v, lane);
#pragma unroll
            for (int k = 0; k < E; ++k) w2[k] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) w2[k] = v[k];
        }
        if (DFT) bfsm::fft_line_step2<N, -1, float>(v, w2, twr, ctx);
        else {
#pragma unroll
            for (int k = 0; k < E; ++k) v[k] = w2[k];
        }
        if (iters > 1) {          // keep the values bounded over many iterations
#pragma unroll
            for (int k = 0; k < E; ++k) { v[k].x *= 0.08838834764f; v[k].y *= 0.08838834764f; }
        }
    }
#pragma unroll
    for (int m = 0; m < E; ++m) out[base + u + TT * m] = v[m];
}

template <int MODE>
double run(const cx<float>* tw, const cx<float>* in, cx<float>* out, int blocks, int iters) {
    const size_t lds = (size_t)N * LS * sizeof(cx<float>);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<MODE><<<blocks, 1024, lds>>>(tw, in, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probe<MODE><<<blocks, 1024, lds>>>(tw, in, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? std::atoi(argv[1]) : 256, iters = argc > 2 ? std::atoi(argv[2]) : 4000;
    std::vector<cx<float>> tw(N), in((size_t)blocks * N * N);
    for (int n = 0; n < N; ++n) tw[n] = {(float)std::cos(-2 * M_PI * n / N), (float)std::sin(-2 * M_PI * n / N)};
    unsigned s = 12345;
    for (auto& z : in) { s = s * 1664525u + 1013904223u; z.x = (float)(s >> 8) / (1 << 24) - 0.5f; s = s * 1664525u + 1013904223u; z.y = (float)(s >> 8) / (1 << 24) - 0.5f; }
    cx<float>*d_tw, *d_in, *d_out;
    CHECK(hipMalloc(&d_tw, N * sizeof(cx<float>)));
    CHECK(hipMalloc(&d_in, in.size() * sizeof(cx<float>)));
    CHECK(hipMalloc(&d_out, in.size() * sizeof(cx<float>)));
    CHECK(hipMemcpy(d_tw, tw.data(), N * sizeof(cx<float>), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_in, in.data(), in.size() * sizeof(cx<float>), hipMemcpyHostToDevice));
    // correctness of both exchanges: one transform of every line of block 0 against a host DFT
    std::vector<cx<float>> got((size_t)N * N);
    for (int mode : {0, 1, 5}) {
        if (mode == 0) run<0>(d_tw, d_in, d_out, blocks, 1); else if (mode == 1) run<1>(d_tw, d_in, d_out, blocks, 1); else run<5>(d_tw, d_in, d_out, blocks, 1);
        CHECK(hipMemcpy(got.data(), d_out, got.size() * sizeof(cx<float>), hipMemcpyDeviceToHost));
        double err = 0, ref = 0;
        for (int line = 0; line < N; line += 7)
            for (int k = 0; k < N; ++k) {
                std::complex<double> acc = 0;
                for (int n = 0; n < N; ++n)
                    acc += std::complex<double>(in[(size_t)line * N + n].x, in[(size_t)line * N + n].y) * std::polar(1.0, -2 * M_PI * n * k / N);
                err = std::max(err, std::abs(acc - std::complex<double>(got[(size_t)line * N + k].x, got[(size_t)line * N + k].y)));
                ref = std::max(ref, std::abs(acc));
            }
        std::printf("mode %d: max error of one 128-point transform vs host DFT: %.2e (relative to max %.2e)\n", mode, err / ref, ref);
        if (!(err / ref < 1e-5)) { std::printf("WRONG RESULT\n"); return 1; }
    }
    const char* names[] = {"LDS exchange + butterflies", "cross-lane exchange + butterflies", "butterflies only", "LDS exchange only", "cross-lane exchange only",
                           "split-plane addtid exchange + butterflies", "split-plane addtid exchange only"};
    double ms[7];
    ms[0] = run<0>(d_tw, d_in, d_out, blocks, iters);
    ms[1] = run<1>(d_tw, d_in, d_out, blocks, iters);
    ms[2] = run<2>(d_tw, d_in, d_out, blocks, iters);
    ms[3] = run<3>(d_tw, d_in, d_out, blocks, iters);
    ms[4] = run<4>(d_tw, d_in, d_out, blocks, iters);
    ms[5] = run<5>(d_tw, d_in, d_out, blocks, iters);
    ms[6] = run<6>(d_tw, d_in, d_out, blocks, iters);
    for (int m = 0; m < 7; ++m)
        std::printf("mode %d  %-36s %8.3f ms  = %7.1f ns per line transform of a 1024-thread workgroup (%d workgroups, %d iterations)\n",
                    m, names[m], ms[m], ms[m] * 1e6 / iters / ((blocks + 255) / 256), blocks, iters);
    return 0;
}
