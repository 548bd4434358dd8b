// bfsm_emu.cpp -- TEST HARNESS ONLY.  Host lock-step emulation of the gfx950 workgroup bodies.
//
// There is no GPU in the authoring container, so the index algebra of the kernels (who owns which point, LDS
// exchange addresses, twiddle indices, layouts, chunk/slab bookkeeping) is unit-tested on the CPU by running the
// SAME bodies (csrc/bfsm_core.hpp) and the SAME plan + launch sequence (csrc/bfsm_pipeline.hpp) with a backend
// in which every GPU thread is a ucontext coroutine and __syncthreads() is a yield to a round-robin scheduler.
// Nothing here is linked into libbfsm_hip.so; the product has no CPU path.
#define BFSM_HD inline __attribute__((always_inline))
#include <ucontext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../boltzmann-fourier-spectral-method_amd/csrc/bfsm_pipeline.hpp"
#include "../../boltzmann-fourier-spectral-method_amd/csrc/bfsm_generic.hpp"

namespace emu {

struct Sched;
struct EmuCtx {
    int tid_, nthreads_, bx_, by_, bz_;
    unsigned char* smem;
    Sched* sched;
    int gx_ = 1, gy_ = 1;
    int gx() const { return gx_; }
    int gy() const { return gy_; }
    int tid() const { return tid_; }
    int nthreads() const { return nthreads_; }
    int bx() const { return bx_; }
    int by() const { return by_; }
    int bz() const { return bz_; }
    int uniform(int v, int) const { return v; }
    template <class U> U* lds() const { return reinterpret_cast<U*>(smem); }
    template <class U> U ldc(const U* p) const { return *p; }
    template <class U> U lds_ld(const U* p) const { return *p; }
    template <class U> void lds_st(U* p, U v) const { *p = v; }
    template <class U> U lds_ld_s(const U* p) const { return *p; }
    template <class U> U ld_stream(const U* p) const { return *p; }
    template <class U> void st_stream(U* p, U v) const { *p = v; }
    template <bool UNI, class U> U ld_stream_at(const U* row, unsigned byte_off) const {
        return *reinterpret_cast<const U*>(reinterpret_cast<const unsigned char*>(row) + byte_off);
    }
    template <bool UNI, class U> U ld_at(const U* row, unsigned byte_off) const { return ld_stream_at<UNI>(row, byte_off); }
    template <bool UNI, class U> void st_at(U* row, unsigned byte_off, U v) const { st_stream_at<UNI>(row, byte_off, v); }
    template <bool UNI, class U> void st_stream_at(U* row, unsigned byte_off, U v) const {
        *reinterpret_cast<U*>(reinterpret_cast<unsigned char*>(row) + byte_off) = v;
    }
    template <bool UNI, class U> void ld_stream_pair_at(const U* row, unsigned byte_off, U& v0, U& v1) const {
        const U* q = reinterpret_cast<const U*>(reinterpret_cast<const unsigned char*>(row) + byte_off);
        v0 = q[0];
        v1 = q[1];
    }
    template <bool UNI, class U> void st_stream_pair_at(U* row, unsigned byte_off, U v0, U v1) const {
        U* q = reinterpret_cast<U*>(reinterpret_cast<unsigned char*>(row) + byte_off);
        q[0] = v0;
        q[1] = v1;
    }
    template <bool UNI, class U> auto ld_real_at(const U* row, unsigned byte_off) const {
        return reinterpret_cast<const U*>(reinterpret_cast<const unsigned char*>(row) + byte_off)->x;
    }
    // cross-lane exchange of DevCtx::xlane_transpose8, through a per-block staging buffer and two wave barriers
    template <class U> void xlane_transpose8(U* v);
    void sync();       // workgroup barrier
    void sched_fence() const {}   // compiler scheduling hint on the device; nothing to do on the host
    void drain_loads() const {}   // wait-count hygiene on the device; nothing to do on the host
    int opaque(int v) const { return v; }
    int opaque_v(int v) const { return v; }
    unsigned lane_off(unsigned v) const { return v; }
    template <class U> U opaque_cx(U v) const { return v; }
    template <class T> void keep_alive(T) const {}
    void wave_sync();  // ordering point inside one wave of 64 threads
};

// Cooperative scheduler.  A thread runs until it reaches a barrier; a workgroup barrier releases when every live
// thread of the block waits at one, a wave barrier when every live thread of that wave waits at one.  Waves are
// advanced one after the other as far as they can go, so code that relies on another WAVE having executed something
// without a workgroup barrier in between reads stale LDS here (the emulation is adversarial on purpose).
struct Sched {
    static constexpr size_t STACK = 256 * 1024;
    enum State : char { RUN = 0, AT_WAVE = 1, AT_BLOCK = 2, DONE = 3 };
    ucontext_t main_ctx;
    std::vector<ucontext_t> ctxs;
    std::vector<char> stacks;
    std::vector<char> state;
    int current = -1;
    bool deadlock = false;
    std::vector<unsigned char> xbuf;     // staging of EmuCtx::xlane_transpose8
    void (*entry)(void*, EmuCtx&) = nullptr;
    void* arg = nullptr;
    std::vector<EmuCtx> ectx;

    static Sched*& active() { static Sched* s = nullptr; return s; }
    static void trampoline() {
        Sched* s = active();
        const int me = s->current;
        s->entry(s->arg, s->ectx[me]);
        s->state[me] = DONE;
        swapcontext(&s->ctxs[me], &s->main_ctx);
    }
    void yield(State st) { const int me = current; state[me] = st; swapcontext(&ctxs[me], &main_ctx); }

    void run_block(int nthreads, int bx, int by, int bz, unsigned char* smem, void (*fn)(void*, EmuCtx&), void* a,
                   int gx = 1, int gy = 1) {
        entry = fn; arg = a;
        if ((int)ctxs.size() < nthreads) { ctxs.resize(nthreads); stacks.resize((size_t)nthreads * STACK); }
        state.assign(nthreads, RUN);
        ectx.resize(nthreads);
        active() = this;
        for (int t = 0; t < nthreads; ++t) {
            ectx[t] = EmuCtx{t, nthreads, bx, by, bz, smem, this, gx, gy};
            getcontext(&ctxs[t]);
            ctxs[t].uc_stack.ss_sp = stacks.data() + (size_t)t * STACK;
            ctxs[t].uc_stack.ss_size = STACK;
            ctxs[t].uc_link = &main_ctx;
            makecontext(&ctxs[t], (void (*)())trampoline, 0);
        }
        const int nwaves = (nthreads + 63) / 64;
        for (;;) {
            bool progressed = false, all_done = true;
            for (int w = 0; w < nwaves; ++w) {
                const int t0 = w * 64, t1 = std::min(nthreads, t0 + 64);
                bool again = true;
                while (again) {          // advance this wave as far as it can go
                    again = false;
                    for (int t = t0; t < t1; ++t) {
                        if (state[t] != RUN) continue;
                        current = t;
                        swapcontext(&main_ctx, &ctxs[t]);
                        progressed = true;
                    }
                    bool at_wave = false, others = false;
                    for (int t = t0; t < t1; ++t) {
                        if (state[t] == AT_WAVE) at_wave = true;
                        else if (state[t] != DONE) others = true;
                    }
                    if (at_wave && !others) {   // whole wave (minus finished lanes) at the wave barrier: release
                        for (int t = t0; t < t1; ++t) if (state[t] == AT_WAVE) state[t] = RUN;
                        again = true;
                    }
                }
            }
            bool at_block = false, others = false;
            for (int t = 0; t < nthreads; ++t) {
                if (state[t] == AT_BLOCK) at_block = true;
                else if (state[t] != DONE) others = true;
                if (state[t] != DONE) all_done = false;
            }
            if (all_done) break;
            if (at_block && !others) {
                for (int t = 0; t < nthreads; ++t) if (state[t] == AT_BLOCK) state[t] = RUN;
                continue;
            }
            if (!progressed) { deadlock = true; break; }   // mismatched barriers
        }
    }
};
inline void EmuCtx::sync() { sched->yield(Sched::AT_BLOCK); }
template <class U> void EmuCtx::xlane_transpose8(U* v) {
    std::vector<unsigned char>& buf = sched->xbuf;
    if (buf.size() < (size_t)nthreads_ * 16 * sizeof(U)) buf.resize((size_t)nthreads_ * 16 * sizeof(U));
    U* b = reinterpret_cast<U*>(buf.data());
    for (int k = 0; k < 16; ++k) b[(size_t)tid_ * 16 + k] = v[k];
    sched->yield(Sched::AT_WAVE);
    const int lane = tid_ & 63, u = lane >> 3, w0 = tid_ - lane + (lane & 7);
    for (int q = 0; q < 2; ++q)
        for (int uu = 0; uu < 8; ++uu) v[q * 8 + uu] = b[(size_t)(w0 + 8 * uu) * 16 + u + 8 * q];
    sched->yield(Sched::AT_WAVE);
}
inline void EmuCtx::wave_sync() { sched->yield(Sched::AT_WAVE); }

struct EmuBackend {
    Sched sched;
    std::vector<unsigned char> smem;
    void* alloc(size_t bytes) { return std::calloc(1, bytes); }
    void release(void* p) { std::free(p); }
    void upload(void* dst, const void* src, size_t bytes) { std::memcpy(dst, src, bytes); }
    void mark(int, double) {}
    bool failed = false;   // a block ended with threads stuck at mismatched barriers

    template <bfsm::K kind, int N, typename T, class P>
    static void body(void* a, EmuCtx& ctx) {
        using namespace bfsm;
        const P& prm = *static_cast<const P*>(a);
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
        else if constexpr (kind == K::GainInvNyq) {
            if constexpr (nyq_rides_along<N>()) body_gain_inv_nyq<N, T>(prm, ctx);
        } else if constexpr (kind == K::GainInvTwo) {
            if constexpr (ab_interleaved<N, T>()) body_gain_inv<N, T, false>(prm, ctx);
        }
    }

    template <bfsm::K kind, int N, typename T, class P>
    void launch_n(int gx, int gy, int gz, const P& prm) {
        const bool pair = kind == bfsm::K::GainInv && bfsm::pair_tile<N>();
        const int threads = pair ? bfsm::pair_threads<N>() : kind == bfsm::K::Reduce ? 256
                            : (bfsm::is_line_kind(kind) ? bfsm::Wg<N>::LINE_THREADS : bfsm::Wg<N>::THREADS);
        smem.assign(pair ? bfsm::pair_lds_bytes<N, T>()
                         : (kind == bfsm::K::GainInv && bfsm::ka_xlane<N, T>()) ? bfsm::ka_xlane_lds_bytes<N, T>()
                         : kind == bfsm::K::GainFwd ? bfsm::kc_lds_bytes<N, T>()
                         : kind == bfsm::K::GainInvNyq ? bfsm::gain_inv_lds_bytes<N, T>()
                         : (bfsm::is_line_kind(kind) ? bfsm::line_lds_bytes<N, T>() : bfsm::tile_lds_bytes<N, T>()), 0xCD);
        P copy = prm;
        for (int bz = 0; bz < gz; ++bz)
            for (int by = 0; by < gy; ++by)
                for (int bx = 0; bx < gx; ++bx) {
                    sched.run_block(threads, bx, by, bz, smem.data(), &body<kind, N, T, P>, &copy, gx, gy);
                    if (sched.deadlock) failed = true;
                }
    }

    template <bfsm::SK kind, typename T, class P>
    static void body_small(void* a, EmuCtx& ctx) {
        const P& prm = *static_cast<const P*>(a);
        if constexpr (kind == bfsm::SK::Gain) bfsm::body_small_gain<T>(prm, ctx);
        else if constexpr (kind == bfsm::SK::Reduce) bfsm::body_small_reduce<T>(prm, ctx);
    }

    template <bfsm::SK kind, typename T, class P>
    void launch_small(int gx, const P& prm) {
        smem.assign(bfsm::small_lds_bytes<T>(), 0xCD);
        P copy = prm;
        for (int bx = 0; bx < gx; ++bx) {
            sched.run_block(bfsm::SMALL_THREADS, bx, 0, 0, smem.data(), &body_small<kind, T, P>, &copy);
            if (sched.deadlock) failed = true;
        }
    }

    template <bfsm::GK kind, typename T, class P>
    static void body_gen(void* a, EmuCtx& ctx) {
        const P& prm = *static_cast<const P*>(a);
        if constexpr (kind == bfsm::GK::Fft) bfsm::body_gen_fft<T, false, bfsm::GEN_C>(prm, ctx);
    else if constexpr (kind == bfsm::GK::FftBig) bfsm::body_gen_fft<T, true, bfsm::GEN_C>(prm, ctx);
    else if constexpr (kind == bfsm::GK::Plane) bfsm::body_gen_plane<T>(prm, ctx);
        else if constexpr (kind == bfsm::GK::Acc) bfsm::body_gen_acc<T>(prm, ctx);
        else if constexpr (kind == bfsm::GK::Combine) bfsm::body_gen_combine<T>(prm, ctx);
        else if constexpr (kind == bfsm::GK::Line3) bfsm::body_gen_line3<T, bfsm::GEN_C>(prm, ctx);
        else if constexpr (kind == bfsm::GK::PlaneAcc) bfsm::body_gen_plane_acc<T>(prm, ctx);
        else if constexpr (kind == bfsm::GK::PlanePair) bfsm::body_gen_plane_pair<T>(prm, ctx);
        else if constexpr (kind == bfsm::GK::Fft8) bfsm::body_gen_fft<T, false, 8>(prm, ctx);
        else if constexpr (kind == bfsm::GK::FftBig8) bfsm::body_gen_fft<T, true, 8>(prm, ctx);
        else if constexpr (kind == bfsm::GK::Line38) bfsm::body_gen_line3<T, 8>(prm, ctx);
    }

    template <bfsm::GK kind, typename T, class P>
    void launch_gen(int gx, int gy, int threads, size_t lds, const P& prm) {
        smem.assign(lds ? lds : 16, 0xCD);
        P copy = prm;
        for (int by = 0; by < gy; ++by)
            for (int bx = 0; bx < gx; ++bx) {
                sched.run_block(threads, bx, by, 0, smem.data(), &body_gen<kind, T, P>, &copy);
                if (sched.deadlock) failed = true;
            }
    }

    template <bfsm::K kind, typename T, class P>
    void launch(int gx, int gy, int gz, const P& prm, int N) {
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
};

// size-generic path (bfsm_generic.hpp): same sequence as the library's entry points for such grids
template <typename T>
int collide_gen_t(const bfsm_desc* d, const double* f, double* Q, double* qhat_out, int nb, bool with_loss) {
    EmuBackend be;
    bfsm::GenericPipeline<T, EmuBackend> p;
    std::string err;
    int rc = p.init(*d, &be, err);
    if (rc) return rc;
    if (nb < 1 || nb > p.max_batch) return BFSM_ERR_INVALID;
    if (p.batch_together()) {                 // as bfsm_collide_batch_partial_async: all members through every launch
        p.gain_partial(f, nb);
        if (qhat_out)
            for (size_t k = 0; k < p.G * (size_t)nb; ++k) { qhat_out[2 * k] = (double)p.qhat[k].x; qhat_out[2 * k + 1] = (double)p.qhat[k].y; }
        if (Q) p.finish(Q, f, with_loss, nb);
        p.destroy();
        return be.failed ? 99 : 0;
    }
    for (int i = 0; i < nb; ++i) {
        p.gain_partial(f + (size_t)i * p.G);
        if (qhat_out)
            for (size_t k = 0; k < p.G; ++k) { qhat_out[2 * ((size_t)i * p.G + k)] = (double)p.qhat[k].x; qhat_out[2 * ((size_t)i * p.G + k) + 1] = (double)p.qhat[k].y; }
        if (Q) p.finish(Q + (size_t)i * p.G, f + (size_t)i * p.G, with_loss);
    }
    p.destroy();
    return be.failed ? 99 : 0;
}

template <typename T>
int collide_t(const bfsm_desc* d, const double* f, double* Q, double* qhat_out, int nb, bool with_loss = true) {
    if (!bfsm::fused_grid(*d)) return collide_gen_t<T>(d, f, Q, qhat_out, nb, with_loss);
    EmuBackend be;
    bfsm::Pipeline<T, EmuBackend> p;
    std::string err;
    int rc = p.init(*d, &be, err);
    if (rc) return rc;
    if (nb < 1 || nb > p.max_batch) return BFSM_ERR_INVALID;
    // qhat requested: the two-call sequence (bfsm_gain_partial, bfsm_finish); otherwise the fused sequence of
    // bfsm_collide / bfsm_collide_batch / bfsm_collide_partial_async (slab reduce inside the first tail kernel)
    const bool fused = qhat_out == nullptr;   // (the library additionally fuses only shards with few slabs)
    if (fused && Q && p.small_path(nb)) {     // N = 16: the whole-direction kernels, as bfsm_collide_partial_async
        p.collide_small(Q, f, with_loss);
        p.destroy();
        return be.failed ? 99 : 0;
    }
    p.gain_partial(f, nb, !fused);
    if (qhat_out) {
        const size_t G = p.plan.G() * (size_t)nb;
        for (size_t i = 0; i < G; ++i) { qhat_out[2 * i] = (double)p.qhat[i].x; qhat_out[2 * i + 1] = (double)p.qhat[i].y; }
    }
    if (Q) p.finish(Q, f, with_loss, nb, fused);
    p.destroy();
    return be.failed ? 99 : 0;
}

// Tail only: f_hat is recomputed (gain of an empty shard), qhat_in replaces the handle's buffer (what the all-reduce
// leaves there on a multi-GPU node), then bfsm_finish.
template <typename T>
int finish_t(const bfsm_desc* d, const double* f, const double* qhat_in, double* Q, int with_loss) {
    EmuBackend be;
    bfsm::Pipeline<T, EmuBackend> p;
    std::string err;
    bfsm_desc e = *d;
    e.dir_begin = e.dir_end = 1;   // empty shard: F1 + zero gain
    int rc = p.init(e, &be, err);
    if (rc) return rc;
    p.gain_partial(f);
    const size_t G = p.plan.G();
    for (size_t i = 0; i < G; ++i) p.qhat[i] = {(T)qhat_in[2 * i], (T)qhat_in[2 * i + 1]};
    p.finish(Q, f, with_loss != 0);
    p.destroy();
    return be.failed ? 99 : 0;
}

template <typename T>
int fft3d_t(int N, double* data, int batch, int sign) {
    EmuBackend be;
    bfsm::Pipeline<T, EmuBackend> p;
    p.be = &be;
    p.plan.N = N;
    std::vector<bfsm::cx<T>> tw(N);
    const long double PI_L = 3.141592653589793238462643383279502884L;
    for (int n = 0; n < N; ++n) {
        const long double a = -2.0L * PI_L * n / N;
        tw[n] = {(T)cosl(a), (T)sinl(a)};
    }
    p.tw = tw.data();
    const size_t total = (size_t)batch * N * N * N;
    std::vector<bfsm::cx<T>> buf(total);
    for (size_t i = 0; i < total; ++i) buf[i] = {(T)data[2 * i], (T)data[2 * i + 1]};
    p.fft3d(buf.data(), batch, sign);
    for (size_t i = 0; i < total; ++i) { data[2 * i] = (double)buf[i].x; data[2 * i + 1] = (double)buf[i].y; }
    p.tw = nullptr;
    return be.failed ? 99 : 0;
}

}  // namespace emu

extern "C" {

// Emulated bfsm_gain_partial + bfsm_finish on host arrays.  qhat_out (optional): 2*G doubles, spectral layout
// [lx][lz][ly].  Q may be NULL to skip the tail.
int bfsm_emu_collide_batch(const bfsm_desc* d, const double* f, double* Q, double* qhat_out, int n_batch) {
    std::string err;
    int rc = bfsm::validate_desc(*d, err);
    if (rc) return rc;
    return d->precision == BFSM_F64 ? emu::collide_t<double>(d, f, Q, qhat_out, n_batch)
                                    : emu::collide_t<float>(d, f, Q, qhat_out, n_batch);
}

// Emulated bfsm_collide_partial_async on a direction shard: fused gain + tail, with or without the loss term.
int bfsm_emu_collide_partial(const bfsm_desc* d, const double* f, double* Q, int with_loss) {
    std::string err;
    int rc = bfsm::validate_desc(*d, err);
    if (rc) return rc;
    return d->precision == BFSM_F64 ? emu::collide_t<double>(d, f, Q, nullptr, 1, with_loss != 0)
                                    : emu::collide_t<float>(d, f, Q, nullptr, 1, with_loss != 0);
}

int bfsm_emu_collide(const bfsm_desc* d, const double* f, double* Q, double* qhat_out) {
    return bfsm_emu_collide_batch(d, f, Q, qhat_out, 1);
}

// Emulated bfsm_finish on a caller-provided (already reduced) Q_gain_hat in the spectral layout.
int bfsm_emu_finish(const bfsm_desc* d, const double* f, const double* qhat_in, double* Q, int with_loss) {
    std::string err;
    int rc = bfsm::validate_desc(*d, err);
    if (rc) return rc;
    return d->precision == BFSM_F64 ? emu::finish_t<double>(d, f, qhat_in, Q, with_loss) : emu::finish_t<float>(d, f, qhat_in, Q, with_loss);
}

// Emulated bfsm_fft3d; data = batch*G interleaved complex doubles (narrowed to float when precision == 32).
int bfsm_emu_fft3d(int N, int precision, double* data, int batch, int sign) {
    if (N != 16 && N != 24 && N != 32 && N != 40 && N != 48 && N != 64 && N != 80 && N != 96 && N != 128) return BFSM_ERR_UNSUPPORTED;
    if (precision == BFSM_F64) return emu::fft3d_t<double>(N, data, batch, sign);
    return emu::fft3d_t<float>(N, data, batch, sign);
}

// Plan introspection for the host-logic tests.  Chunk rows: (n_seg, dir0, n, per_group, seg0).  Segment rows:
// (chunk, d0 relative to the chunk, n, r, 0).  Returns the chunk count; *n_segs receives the segment count.
int bfsm_emu_plan(const bfsm_desc* d, int* chunk_rows, int max_chunks, int* seg_rows, int max_segs, int* n_segs) {
    std::string err;
    int rc = bfsm::validate_desc(*d, err);
    if (rc) return -rc;
    bfsm::PlanInfo p = bfsm::make_plan(*d);
    if (n_segs) *n_segs = (int)p.segs.size();
    int n = 0;
    for (const auto& c : p.chunks) {
        if (n < max_chunks) {
            int* r = chunk_rows + 5 * n;
            r[0] = c.n_seg; r[1] = (int)c.dir0; r[2] = c.n; r[3] = c.per_group; r[4] = c.seg0;
        }
        for (int i = c.seg0; i < c.seg0 + c.n_seg && i < max_segs; ++i) {
            int* r = seg_rows + 5 * i;
            r[0] = n; r[1] = p.segs[i].d0; r[2] = p.segs[i].n; r[3] = p.segs[i].r; r[4] = 0;
        }
        ++n;
    }
    return n;
}
}
