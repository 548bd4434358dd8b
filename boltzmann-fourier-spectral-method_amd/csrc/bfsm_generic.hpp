// bfsm_generic.hpp -- size-generic path of the collision operator: any even Nvx, Nvy, Nvz <= 256 whose prime factors
// are 2, 3, 5, 7, 11, 13 (non-cubic boxes, N = 20, 112, 160, ...), which the reference plans with cufftPlan3d / cufftPlanMany
// (Collisions/CUDABoltzmannOperator.cu:86-100) and fftw_plan_dft_3d (Collisions/FFTWBoltzmannOperator.cpp:64-65).
//
// The cubic grids N in {16, 24, 32, 40, 48, 64, 80, 96, 128} run on the fused three-kernel pipeline of bfsm_core.hpp (6 array passes
// per direction); everything else runs here, on mixed-radix (8, 4, 2, 3, 5; 7, 11, 13 table-driven) Stockham passes in LDS
// with runtime sizes.  Three sequences, chosen per box in GenericPipeline (profiles/r04_generic_fused_ab.txt):
//   * the (y,z) plane fits the LDS (most boxes up to ~1300 plane points in double precision): the cubic pipeline's three
//     kernels in size-generic form -- body_gen_plane_pair / body_gen_plane straight from f_hat with the phase multiply on the
//     load side (compute_alpha_times_f_hat, BoltzmannCUDAKernels.cu:21-59), body_gen_line3 (x part of both inverse transforms,
//     hadamard_product Kernels.cu:62-74, x part of the forward transform), body_gen_plane_acc ((y,z) forward transform +
//     atomic-free accumulate, Kernels.cu:79-123) -- 6 array moves per direction, 2.5-2.9 TB/s algorithmic; batches of
//     distributions go through these launches together;
//   * bigger planes: per-axis passes with the x-line kernel in the middle, 14 moves (8 where only the x lines are too long
//     for the first form);
//   * x axes with a factor 7, 11 or 13: one pass per axis (or x + a plane pass), 18 / 12 moves.
// The other pointwise steps ride on load sides as before: beta2 * f_hat (Kernels.cu:126-159) on the first tail pass,
// copy_to_complex (Kernels.cu:4-16) on the first pass of FFT(f).  Layouts are the reference's own: physical [x][y][z],
// spectral [lx][ly][lz], z contiguous.
//
// Like bfsm_core.hpp the kernel bodies are templates over an execution context, so tests/emu runs the same code on the
// host.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "bfsm_pipeline.hpp"

namespace bfsm {

constexpr int GEN_MAX_N = 256;
constexpr int GEN_THREADS = 256;

// load-side fusions of a pass
enum : int { GEN_PLAIN = 0, GEN_PHASE = 1, GEN_PRODUCT = 2, GEN_REAL = 3, GEN_BETA2 = 4, GEN_TAIL2 = 5 };
// GEN_TAIL2 (x pass of the tail): slot 0 of a member transforms in (Q_hat) as it is, slot 1 transforms beta2 * in2 (f_hat) -- the
// gain and the loss term in one launch

template <typename T>
struct GenFftParams {
    const void* in;          // cx<T> arrays [batch][G]; GEN_REAL: const double* [batch][G]; GEN_PHASE / GEN_BETA2: f_hat [G]
    const cx<T>* in2;        // GEN_PRODUCT: second factor, same shape as in
    cx<T>* out;              // [batch][G]
    const cx<T>* tw;         // exp(-2 pi i k / n), k < n, n = length of the transformed axis
    int nx, ny, nz;
    int axis;                // 0: x, 1: y, 2: z
    int sign;                // -1 forward, +1 backward (FFTW convention), unnormalised
    int C;                   // lines ("columns") per workgroup
    int n_radix;
    int radix[8];
    int mode;
    // GEN_PHASE: batch member b = 2 * d + s of the chunk; factor phx[d][lx] phy[d][ly] phz[d][lz] (phx carries 1/G),
    // conjugated for s = 1 (alpha2 = conj(alpha1), FFTWBoltzmannOperator.cpp:219-224)
    const cx<T>* phx;
    const cx<T>* phy;
    const cx<T>* phz;
    long long dir0;
    // GEN_BETA2: beta2[|l|^2] (1/G folded)
    const T* beta2;
    size_t in_bstride;       // elements between batch members of in / in2 (0: shared)
    size_t out_bstride;      // elements between batch members of out
    // plane kernel (body_gen_plane): the y and the z pass of one x-plane in one workgroup; tw / radix above describe the
    // FIRST axis transformed (z in the forward direction, y in the backward one), these the second
    const cx<T>* tw_b;
    int n_radix_b;
    int radix_b[8];
    // batches of distributions evaluated together: grid.y = members x mper, member m at + m * in_mstride / out_mstride and the
    // index inside the member (what the fusions above call b) = grid.y index mod mper.  mper = 0: no member level.
    int mper;
    size_t in_mstride, out_mstride;
};

template <typename T>
struct GenAccParams {        // Q_hat[l] (+)= sum_d dirw[d] beta1[r(d)][|l|^2] P_hat[d][l], fixed order (no atomics)
    const cx<T>* p;          // member d at p + d * p_bstride
    size_t p_bstride;
    cx<T>* qhat;             // [G]
    const T* dirw;           // [shard directions]
    const int* rdir;         // [shard directions] radial node of each direction
    const T* beta1;          // [n_gl][n2stride]
    long long dir0;
    int n;
    int n2stride;
    int first;               // != 0: start from zero instead of from qhat
    int nx, ny, nz;
    size_t p_mstride, q_mstride;   // grid.y = members
};

// Fused sequence of the boxes whose (y,z) plane fits the LDS (round 4): the three kernels of the cubic pipeline in
// size-generic form -- 6 array moves per direction instead of 12.
template <typename T>
struct GenLineParams {       // x part of both inverse transforms + product + x part of the forward transform
    cx<T>* a;                // [2 n][G]: A1', A2' of direction d at members 2d, 2d + 1 ([lx][y][z]); P' overwrites member 2d
    const cx<T>* tw;         // x-axis twiddles
    int nx, ny, nz;
    int n_radix;
    int radix[8];
    int n;                   // directions of the chunk: grid.y = members x n
    size_t mstride;          // elements between the members' scratch
};

template <typename T>
struct GenPlaneAccParams {   // (y,z) part of the forward transform + weighted sum over a group of directions
    const cx<T>* p;          // P' of direction d at p + d * p_bstride
    size_t p_bstride;
    cx<T>* slab;             // [groups][G]: sum over the group's directions of dirw beta1 P_hat
    const T* dirw;
    const int* rdir;
    const T* beta1;
    long long dir0;          // shard index of the chunk's first direction
    int n;                   // directions of the chunk
    int per_group;
    int n2stride;
    int nx, ny, nz;
    const cx<T>* tw;         // z axis (transformed first), then y
    const cx<T>* tw_b;
    int n_radix, n_radix_b;
    int radix[8], radix_b[8];
    int groups;              // grid.y = members x groups
    size_t p_mstride;        // elements between the members' scratch; member m's slabs at slab + m * groups * G
};

template <typename T>
struct GenCombineParams {    // Q = Re(gain) - Re(loss) * f   (compute_Q_total, Kernels.cu:162-177)
    const cx<T>* g;
    const cx<T>* l;
    const double* f;
    double* Q;
    size_t G;
    int with_loss;
};

// floor(w / d) for 0 <= w < 65536, 1 <= d <= 256 from the reciprocal inv = 1.0f / d: five instructions instead of the thirty of
// an integer division by a run-time value.  Exact: the fractional part of (w + 0.5) / d lies in [0.5 / d, 1 - 0.5 / d] and the
// rounding error of the product is below 0.008 / d.
BFSM_HD int gen_div(int w, float inv) { return (int)(((float)w + 0.5f) * inv); }

// Fourier mode of index i on an axis of n points (FFTWBoltzmannOperator.cpp:50-57)
BFSM_HD int gen_mode(int i, int n) { return i < n / 2 ? i : i - n; }

template <int R, typename T>
BFSM_HD void gen_dft(cx<T>* x, int sgn) {
    // direct small DFTs, X[k] = sum_j x[j] exp(sgn * 2 pi i j k / R)
    if constexpr (R == 2) {
        const cx<T> a = x[0], b = x[1];
        x[0] = cadd(a, b);
        x[1] = csub(a, b);
    } else if constexpr (R == 4) {
        const cx<T> a = cadd(x[0], x[2]), b = csub(x[0], x[2]), c = cadd(x[1], x[3]), d = csub(x[1], x[3]);
        const cx<T> id = sgn > 0 ? cx<T>{-d.y, d.x} : cx<T>{d.y, -d.x};     // sgn * i * d
        x[0] = cadd(a, c);
        x[2] = csub(a, c);
        x[1] = cadd(b, id);
        x[3] = csub(b, id);
    } else if constexpr (R == 8) {
        // two radix-4 transforms of the even / odd inputs, then X[k], X[k + 4] = E[k] +- w^k O[k], w = exp(sgn 2 pi i / 8)
        cx<T> e[4] = {x[0], x[2], x[4], x[6]}, o[4] = {x[1], x[3], x[5], x[7]};
        gen_dft<4, T>(e, sgn);
        gen_dft<4, T>(o, sgn);
        const T h = (T)0.70710678118654752440;
        const cx<T> o1 = sgn > 0 ? cx<T>{h * (o[1].x - o[1].y), h * (o[1].x + o[1].y)} : cx<T>{h * (o[1].x + o[1].y), h * (o[1].y - o[1].x)};
        const cx<T> o2 = sgn > 0 ? cx<T>{-o[2].y, o[2].x} : cx<T>{o[2].y, -o[2].x};
        const cx<T> o3 = sgn > 0 ? cx<T>{-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y)} : cx<T>{h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y)};
        x[0] = cadd(e[0], o[0]); x[4] = csub(e[0], o[0]);
        x[1] = cadd(e[1], o1);   x[5] = csub(e[1], o1);
        x[2] = cadd(e[2], o2);   x[6] = csub(e[2], o2);
        x[3] = cadd(e[3], o3);   x[7] = csub(e[3], o3);
    } else if constexpr (R == 3) {
        const T c = (T)-0.5, s = (T)(sgn * 0.86602540378443864676);
        const cx<T> t1 = cadd(x[1], x[2]), t2 = csub(x[1], x[2]);
        const cx<T> m = {x[0].x + c * t1.x, x[0].y + c * t1.y};
        const cx<T> r = {-s * t2.y, s * t2.x};                               // i * s * t2
        x[0] = cadd(x[0], t1);
        x[1] = cadd(m, r);
        x[2] = csub(m, r);
    } else {   // R == 5
        const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;
        const T s1 = (T)(sgn * 0.95105651629515357212), s2 = (T)(sgn * 0.58778525229247312917);
        const cx<T> a1 = cadd(x[1], x[4]), b1 = csub(x[1], x[4]), a2 = cadd(x[2], x[3]), b2 = csub(x[2], x[3]);
        const cx<T> m1 = {x[0].x + c1 * a1.x + c2 * a2.x, x[0].y + c1 * a1.y + c2 * a2.y};
        const cx<T> m2 = {x[0].x + c2 * a1.x + c1 * a2.x, x[0].y + c2 * a1.y + c1 * a2.y};
        const cx<T> r1 = {-(s1 * b1.y + s2 * b2.y), s1 * b1.x + s2 * b2.x};  // i * (s1 b1 + s2 b2)
        const cx<T> r2 = {-(s2 * b1.y - s1 * b2.y), s2 * b1.x - s1 * b2.x};  // i * (s2 b1 - s1 b2)
        x[0] = cadd(x[0], cadd(a1, a2));
        x[1] = cadd(m1, r1);
        x[4] = csub(m1, r1);
        x[2] = cadd(m2, r2);
        x[3] = csub(m2, r2);
    }
}

// Odd prime radices without a hand-written butterfly (7, 11, 13): direct R x R transform with the roots of unity read
// from the axis' own twiddle table (R divides n, so exp(-2 pi i m / R) = tw[m n / R]).  O(R^2) per point group: sizes
// with these factors are served, not tuned.
template <int R, typename T>
BFSM_HD void gen_dft_table(cx<T>* x, int sgn, const cx<T>* tw, int n) {
    cx<T> y[R];
    const int step = n / R;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        cx<T> acc = x[0];
#pragma unroll
        for (int j = 1; j < R; ++j) {
            const cx<T> w = tw[((j * k) % R) * step];
            const cx<T> t = sgn < 0 ? cmul(x[j], w) : cmulc(x[j], w);
            acc = cadd(acc, t);
        }
        y[k] = acc;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) x[k] = y[k];
}

// Lines per workgroup and thread geometry of a pass: 256 threads = 16 lines x 16 "rows"; thread (row, line) walks the
// points row, row + 16, ... of its line, so no per-element integer division is ever needed.
constexpr int GEN_C = 16;
constexpr int GEN_LS = GEN_C + 1;

// One Stockham pass of radix R over GEN_C lines of n points held in LDS as [point][line] (row stride GEN_LS).
// ns = product of the radices already applied; NS_POW2: ns is a power of two (shift / mask instead of division: the
// plan orders the radices 8, 4, 2 first and 3, 5 last, so only passes behind an odd radix take the general form).
// The axis' twiddle table copied into LDS next to the line buffers: a pass reads R - 1 table values per butterfly, and a
// global (cached) load costs a pass most of a microsecond of latency that nothing hides in workgroups this small
// (round 4: the y/z plane kernel of 32 x 64 x 16 took 10 us per workgroup, half of it waiting for twiddles).
template <typename T, class Ctx>
BFSM_HD void gen_stage_tw(cx<T>* dst, const cx<T>* tw, int n, Ctx& ctx) {
    for (int i = ctx.tid(); i < n; i += GEN_THREADS) dst[i] = tw[i];
}

// FIRST: the pass with ns = 1, whose twiddles are all 1 (k = 0): no table reads, no multiplications.
template <int R, bool NS_POW2, bool FIRST, int C, typename T, class Ctx>
BFSM_HD void gen_pass_impl(const cx<T>* src, cx<T>* dst, const cx<T>* tw, int n, int ns, int sgn, Ctx& ctx) {
    const int m = n / R;
    const int col = ctx.tid() % C, row = ctx.tid() / C;
    const int tstep = n / (ns * R);                    // twiddle exp(sgn 2 pi i k q / (ns R)) = tw[k q tstep], k q tstep < n
    const float inv_ns = 1.0f / (float)ns;
    const int sh = NS_POW2 ? (31 - __builtin_clz((unsigned)ns)) : 0;
    for (int j = row; j < m; j += GEN_THREADS / C) {
        const int hi = NS_POW2 ? (j >> sh) : gen_div(j, inv_ns);
        const int k = NS_POW2 ? (j & (ns - 1)) : (j - hi * ns);   // position inside the sub-transform done so far
        cx<T> x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            x[q] = src[(j + q * m) * (C + 1) + col];
            if (!FIRST && q > 0) {
                const cx<T> w = tw[k * q * tstep];
                x[q] = sgn < 0 ? cmul(x[q], w) : cmulc(x[q], w);
            }
        }
        if constexpr (R == 7 || R == 11 || R == 13) gen_dft_table<R, T>(x, sgn, tw, n);
        else gen_dft<R, T>(x, sgn);
        const int j0 = hi * ns * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) dst[(j0 + q * ns) * (C + 1) + col] = x[q];
    }
}
template <int R, bool NS_POW2, int C, typename T, class Ctx>
BFSM_HD void gen_pass(const cx<T>* src, cx<T>* dst, const cx<T>* tw, int n, int ns, int sgn, Ctx& ctx) {
    // (the twiddle-free instantiation pays in the plane kernels only: with it the x-line kernel measured 7 % slower)
    gen_pass_impl<R, NS_POW2, false, C, T>(src, dst, tw, n, ns, sgn, ctx);
}

// Batched 1-D transform along one axis.  grid = (blocks of C lines, batch).  Workgroup: GEN_THREADS threads, LDS =
// two buffers of n x (C + 1) complex + the axis' n twiddles.  A "line" is the set of n points along the transformed axis; consecutive
// lines are consecutive in z (axes x, y) or consecutive (x, y) pairs (axis z).
// BIG: the instantiation that also carries the table-driven radix-7 / 11 / 13 butterflies (13 complex inputs + 13 outputs
// in registers: 256 VGPRs and a kilobyte of scratch per lane in double precision).  Axes whose factors are 2, 3, 5 only --
// nearly every box -- run the instantiation without them (round 4; GK::Fft / GK::FftBig).
template <typename T, bool BIG, int C, class Ctx>
BFSM_HD void body_gen_fft(const GenFftParams<T>& prm, Ctx& ctx) {
    const int nx = prm.nx, ny = prm.ny, nz = prm.nz;
    const int axis = prm.axis;
    const int n = axis == 0 ? nx : (axis == 1 ? ny : nz);
    const size_t G = (size_t)nx * ny * nz;
    const int ncols = (int)(G / (size_t)n);
    int b = ctx.by();
    size_t in_moff = 0, out_moff = 0;
    if (prm.mper > 0) {
        const int m = b / prm.mper;
        b -= m * prm.mper;
        in_moff = (size_t)m * prm.in_mstride;
        out_moff = (size_t)m * prm.out_mstride;
    }
    cx<T>* buf0 = ctx.template lds<cx<T>>();
    cx<T>* buf1 = buf0 + (size_t)n * (C + 1);
    cx<T>* twl = buf1 + (size_t)n * (C + 1);
    gen_stage_tw<T>(twl, prm.tw, n, ctx);
    // Global <-> LDS: lanes run along the contiguous (z) direction of memory.  Axes x, y: C consecutive lanes take the C
    // lines of the block at one point; axis z: 256 / C consecutive lanes take as many consecutive points of one line.
    constexpr int PW = GEN_THREADS / C;                // points covered per step
    const int cl = axis == 2 ? ctx.tid() / PW : ctx.tid() % C;    // line of the block this thread moves
    const int p0 = axis == 2 ? ctx.tid() % PW : ctx.tid() / C;    // first point; then steps of PW
    const int col = ctx.bx() * C + cl;
    const bool live = col < ncols;
    // offset of point 0 of the line and the stride between its points; (ix, iy, iz) of point 0 for the fused factors
    size_t base = 0, ps = 1;
    int i0x = 0, i0y = 0, i0z = 0;
    if (axis == 0) { base = (size_t)col; ps = (size_t)ny * nz; i0y = col / nz; i0z = col - i0y * nz; }
    else if (axis == 1) { i0x = col / nz; i0z = col - i0x * nz; base = (size_t)i0x * ny * nz + i0z; ps = (size_t)nz; }
    else { i0x = col / ny; i0y = col - i0x * ny; base = (size_t)col * nz; ps = 1; }
    const size_t in_off = in_moff + (size_t)b * prm.in_bstride;
    // GEN_PHASE: the factors of the two axes that are NOT transformed are constant along the line: one product per thread,
    // then one table value and two complex multiplications per point instead of three and three
    cx<T> ph_line = {(T)1, (T)0};
    if (prm.mode == GEN_PHASE && live) {
        const size_t d = (size_t)(prm.dir0 + (b >> 1));
        const cx<T> px = prm.phx[d * nx + i0x], py = prm.phy[d * ny + i0y], pz = prm.phz[d * nz + i0z];
        ph_line = axis == 0 ? cmul(py, pz) : (axis == 1 ? cmul(px, pz) : cmul(px, py));
    }
    const bool loss_slot = prm.mode == GEN_TAIL2 && (b & 1);
    const cx<T>* inc = loss_slot ? prm.in2 : static_cast<const cx<T>*>(prm.in);
    // the global loads are issued four points at a time before any of them is consumed: the trip count is a run-time
    // value, and a rolled loop would pay one full memory latency per point
    constexpr int STEP = GEN_THREADS / C, CH = 4;
    for (int pc = p0; pc < n; pc += CH * STEP) {
        cx<T> vin[CH], vin2[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int pt = pc + i * STEP;
            vin[i] = {(T)0, (T)0};
            vin2[i] = {(T)1, (T)0};
            if (live && pt < n) {
                const size_t idx = base + (size_t)pt * ps;
                if (prm.mode == GEN_REAL) vin[i].x = (T) static_cast<const double*>(prm.in)[in_off + idx];
                else vin[i] = inc[in_off + idx];
                if (prm.mode == GEN_PRODUCT) vin2[i] = prm.in2[in_off + idx];
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int pt = pc + i * STEP;
            if (pt < n) {
                cx<T> v = vin[i];
                if (live) {
                    if (prm.mode == GEN_PRODUCT) {
                        v = cmul(v, vin2[i]);
                    } else if (prm.mode == GEN_PHASE || prm.mode == GEN_BETA2 || loss_slot) {
                        const int ix = axis == 0 ? pt : i0x, iy = axis == 1 ? pt : i0y, iz = axis == 2 ? pt : i0z;
                        if (prm.mode == GEN_PHASE) {
                            const size_t d = (size_t)(prm.dir0 + (b >> 1));
                            const cx<T> pa = axis == 0 ? prm.phx[d * nx + ix] : (axis == 1 ? prm.phy[d * ny + iy] : prm.phz[d * nz + iz]);
                            const cx<T> ph = cmul(pa, ph_line);
                            v = (b & 1) ? cmulc(v, ph) : cmul(v, ph);
                        } else {
                            const int mx = gen_mode(ix, nx), my = gen_mode(iy, ny), mz = gen_mode(iz, nz);
                            const T b2 = prm.beta2[mx * mx + my * my + mz * mz];
                            v = {b2 * v.x, b2 * v.y};
                        }
                    }
                }
                buf0[pt * (C + 1) + cl] = v;
            }
        }
    }
    ctx.sync();
    cx<T>* src = buf0;
    cx<T>* dst = buf1;
    int ns = 1;
    for (int r = 0; r < prm.n_radix; ++r) {
        const int R = prm.radix[r];
        const bool p2 = (ns & (ns - 1)) == 0;
        if (R == 8) gen_pass<8, true, C, T>(src, dst, twl, n, ns, prm.sign, ctx);          // radices 8, 4, 2 come first
        else if (R == 4) gen_pass<4, true, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
        else if (R == 2) gen_pass<2, true, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
        else if (R == 3 && p2) gen_pass<3, true, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
        else if (R == 3) gen_pass<3, false, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
        else if (R == 5 && p2) gen_pass<5, true, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
        else if (R == 5) gen_pass<5, false, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
        else if constexpr (BIG) {
            if (R == 7) gen_pass<7, false, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
            else if (R == 11) gen_pass<11, false, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
            else gen_pass<13, false, C, T>(src, dst, twl, n, ns, prm.sign, ctx);
        }
        ns *= R;
        ctx.sync();
        cx<T>* t = src; src = dst; dst = t;
    }
    if (live) {
        const size_t out_off = out_moff + (size_t)b * prm.out_bstride;
        for (int pt = p0; pt < n; pt += GEN_THREADS / C) prm.out[out_off + base + (size_t)pt * ps] = src[pt * (C + 1) + cl];
    }
}

// One Stockham pass of radix R over ALL nl lines of n points of a plane held in LDS; element (point pt, line l) at
// pt * ps + l * ls.  The nl * n / R butterflies are dealt to the threads as one flat range, lanes along the lines (unit or
// odd stride in LDS either way), so that every thread works whenever the plane has 256 butterflies to give.
template <int R, bool NS_POW2, bool FIRST, typename T, class Ctx>
BFSM_HD void gen_plane_pass_impl(const cx<T>* src, cx<T>* dst, const cx<T>* tw, int n, int ns, int sgn, int nl, int ps, int ls, Ctx& ctx) {
    const int m = n / R;
    const int tstep = n / (ns * R);
    const int sh = NS_POW2 ? (31 - __builtin_clz((unsigned)ns)) : 0;
    const int total = nl * m;                          // < 65536: a plane of the plane kernels has at most 2560 points
    const float inv_nl = 1.0f / (float)nl, inv_ns = 1.0f / (float)ns;
    for (int w = ctx.tid(); w < total; w += GEN_THREADS) {
        const int j = gen_div(w, inv_nl), l = w - j * nl;
        const cx<T>* s0 = src + (size_t)l * ls;
        cx<T>* d0 = dst + (size_t)l * ls;
        const int hi = NS_POW2 ? (j >> sh) : gen_div(j, inv_ns);
        const int k = NS_POW2 ? (j & (ns - 1)) : (j - hi * ns);
        cx<T> x[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            x[q] = s0[(j + q * m) * ps];
            if (!FIRST && q > 0) {
                const cx<T> wq = tw[k * q * tstep];
                x[q] = sgn < 0 ? cmul(x[q], wq) : cmulc(x[q], wq);
            }
        }
        gen_dft<R, T>(x, sgn);
        const int j0 = hi * ns * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) d0[(j0 + q * ns) * ps] = x[q];
    }
}
template <int R, bool NS_POW2, typename T, class Ctx>
BFSM_HD void gen_plane_pass(const cx<T>* src, cx<T>* dst, const cx<T>* tw, int n, int ns, int sgn, int nl, int ps, int ls, Ctx& ctx) {
    if (ns == 1) gen_plane_pass_impl<R, true, true, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
    else gen_plane_pass_impl<R, NS_POW2, false, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
}

template <typename T, class Ctx>
BFSM_HD void gen_plane_axis(cx<T>*& src, cx<T>*& dst, const cx<T>* tw, const int* radix, int n_radix, int n, int sgn, int nl, int ps,
                            int ls, Ctx& ctx) {
    int ns = 1;
    for (int r = 0; r < n_radix; ++r) {
        const int R = radix[r];
        const bool p2 = (ns & (ns - 1)) == 0;
        if (R == 8) gen_plane_pass<8, true, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
        else if (R == 4) gen_plane_pass<4, true, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
        else if (R == 2) gen_plane_pass<2, true, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
        else if (R == 3 && p2) gen_plane_pass<3, true, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
        else if (R == 3) gen_plane_pass<3, false, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
        else if (R == 5 && p2) gen_plane_pass<5, true, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
        else gen_plane_pass<5, false, T>(src, dst, tw, n, ns, sgn, nl, ps, ls, ctx);
        ns *= R;
        ctx.sync();
        cx<T>* t = src; src = dst; dst = t;
    }
}

// The y and the z pass of ONE x-plane fused through LDS (round 4; boxes whose plane fits: 2 ny (nz + 1) elements of LDS,
// radices 2, 3, 5): the plane [y][z] is contiguous in memory, so the loads and stores are fully coalesced, and a 3-D
// transform is 2 array passes here + 1 x-pass instead of 3 passes -- 12 array moves per direction instead of 18.
// grid = (nx planes, batch).  Forward (sign -1): z then y; backward: y then z.  Same load-side fusions as body_gen_fft.
template <typename T, class Ctx>
BFSM_HD void body_gen_plane(const GenFftParams<T>& prm, Ctx& ctx) {
    const int nx = prm.nx, ny = prm.ny, nz = prm.nz;
    const int ix = ctx.bx();
    int b = ctx.by();
    size_t in_moff = 0, out_moff = 0;
    if (prm.mper > 0) {
        const int m = b / prm.mper;
        b -= m * prm.mper;
        in_moff = (size_t)m * prm.in_mstride;
        out_moff = (size_t)m * prm.out_mstride;
    }
    const int LSZ = nz + 1;                                  // LDS row stride of the plane (odd: conflict-free both ways)
    cx<T>* buf0 = ctx.template lds<cx<T>>();
    cx<T>* buf1 = buf0 + (size_t)ny * LSZ;
    cx<T>* twa = buf1 + (size_t)ny * LSZ;                    // twiddles of the first axis transformed, then of the second
    cx<T>* twb = twa + (prm.sign < 0 ? nz : ny);
    gen_stage_tw<T>(twa, prm.tw, prm.sign < 0 ? nz : ny, ctx);
    gen_stage_tw<T>(twb, prm.tw_b, prm.sign < 0 ? ny : nz, ctx);
    const size_t plane = (size_t)ny * nz, base = (size_t)ix * plane;
    const size_t in_off = in_moff + (size_t)b * prm.in_bstride;
    // (iy, iz) of the element this thread touches, advanced by GEN_THREADS elements per step without a division
    const int dy = GEN_THREADS / nz, dz = GEN_THREADS - dy * nz;
    int iy = ctx.tid() / nz, iz = ctx.tid() - iy * nz;
    for (size_t e = (size_t)ctx.tid(); e < plane; e += GEN_THREADS) {
        const size_t idx = base + e;
        cx<T> v = {(T)0, (T)0};
        if (prm.mode == GEN_REAL) v.x = (T) static_cast<const double*>(prm.in)[in_off + idx];
        else v = static_cast<const cx<T>*>(prm.in)[in_off + idx];
        if (prm.mode == GEN_PRODUCT) v = cmul(v, prm.in2[in_off + idx]);
        else if (prm.mode == GEN_PHASE) {
            const size_t d = (size_t)(prm.dir0 + (b >> 1));
            const cx<T> ph = cmul(cmul(prm.phx[d * nx + ix], prm.phy[d * ny + iy]), prm.phz[d * nz + iz]);
            v = (b & 1) ? cmulc(v, ph) : cmul(v, ph);
        } else if (prm.mode == GEN_BETA2) {
            const int mx = gen_mode(ix, nx), my = gen_mode(iy, ny), mz = gen_mode(iz, nz);
            const T b2 = prm.beta2[mx * mx + my * my + mz * mz];
            v = {b2 * v.x, b2 * v.y};
        }
        buf0[iy * LSZ + iz] = v;
        iy += dy; iz += dz;
        if (iz >= nz) { iz -= nz; ++iy; }
    }
    ctx.sync();
    cx<T>* src = buf0;
    cx<T>* dst = buf1;
    // first axis: z when transforming forward (lines = y rows, points contiguous), y when transforming backward
    if (prm.sign < 0) {
        gen_plane_axis<T>(src, dst, twa, prm.radix, prm.n_radix, nz, prm.sign, ny, 1, LSZ, ctx);
        gen_plane_axis<T>(src, dst, twb, prm.radix_b, prm.n_radix_b, ny, prm.sign, nz, LSZ, 1, ctx);
    } else {
        gen_plane_axis<T>(src, dst, twa, prm.radix, prm.n_radix, ny, prm.sign, nz, LSZ, 1, ctx);
        gen_plane_axis<T>(src, dst, twb, prm.radix_b, prm.n_radix_b, nz, prm.sign, ny, 1, LSZ, ctx);
    }
    const size_t out_off = out_moff + (size_t)b * prm.out_bstride;
    iy = ctx.tid() / nz; iz = ctx.tid() - iy * nz;
    for (size_t e = (size_t)ctx.tid(); e < plane; e += GEN_THREADS) {
        prm.out[out_off + base + e] = src[iy * LSZ + iz];
        iy += dy; iz += dz;
        if (iz >= nz) { iz -= nz; ++iy; }
    }
}

// All Stockham passes of one axis over the C lines of a block held in LDS (the loop of body_gen_fft, radices 2..5)
template <typename T, int C, class Ctx>
BFSM_HD void gen_line_axis(cx<T>*& src, cx<T>*& dst, const cx<T>* tw, const int* radix, int n_radix, int n, int sgn, Ctx& ctx) {
    int ns = 1;
    for (int r = 0; r < n_radix; ++r) {
        const int R = radix[r];
        const bool p2 = (ns & (ns - 1)) == 0;
        if (R == 8) gen_pass<8, true, C, T>(src, dst, tw, n, ns, sgn, ctx);
        else if (R == 4) gen_pass<4, true, C, T>(src, dst, tw, n, ns, sgn, ctx);
        else if (R == 2) gen_pass<2, true, C, T>(src, dst, tw, n, ns, sgn, ctx);
        else if (R == 3 && p2) gen_pass<3, true, C, T>(src, dst, tw, n, ns, sgn, ctx);
        else if (R == 3) gen_pass<3, false, C, T>(src, dst, tw, n, ns, sgn, ctx);
        else if (R == 5 && p2) gen_pass<5, true, C, T>(src, dst, tw, n, ns, sgn, ctx);
        else gen_pass<5, false, C, T>(src, dst, tw, n, ns, sgn, ctx);
        ns *= R;
        ctx.sync();
        cx<T>* t = src; src = dst; dst = t;
    }
}

// x-lines of one direction: A1 = IFFT_x(A1'), A2 = IFFT_x(A2'), P = A1 * A2 (hadamard_product, Kernels.cu:62-74),
// P' = FFT_x(P), written over A1' -- what KB does on the cubes.  grid = (blocks of C lines, directions); LDS = three
// buffers of nx x (C + 1): the second transform ping-pongs between the third buffer and the one the first left free.
template <typename T, int C, class Ctx>
BFSM_HD void body_gen_line3(const GenLineParams<T>& prm, Ctx& ctx) {
    const int n = prm.nx;
    const size_t ps = (size_t)prm.ny * prm.nz, G = ps * (size_t)n;
    cx<T>* X = ctx.template lds<cx<T>>();
    cx<T>* Y = X + (size_t)n * (C + 1);
    cx<T>* Z = Y + (size_t)n * (C + 1);
    cx<T>* twl = Z + (size_t)n * (C + 1);
    if constexpr (C != 8) gen_stage_tw<T>(twl, prm.tw, n, ctx);      // (the 8-line form has its own layout, below)
    const int cl = ctx.tid() % C, p0 = ctx.tid() / C;
    const size_t col = (size_t)ctx.bx() * C + cl;
    const bool live = col < ps;
    const int mem = ctx.by() / prm.n, dl = ctx.by() - mem * prm.n;
    cx<T>* A1 = prm.a + (size_t)mem * prm.mstride + (size_t)dl * 2 * G + col;
    const cx<T>* A2 = A1 + G;
    constexpr int STEP = GEN_THREADS / C, CH = 4;
    if constexpr (C == 8) {
        // Long lines (the 8-line form): TWO buffers.  A thread owns n / 32 <= 8 points of its line; A2' waits in registers while
        // A1' is transformed, the transformed A1 waits in registers while A2' is: 49 KB instead of 72 KB of LDS at n = 160,
        // three workgroups per CU instead of two.
        constexpr int K = GEN_MAX_N / STEP;
        cx<T> k1[K], k2[K];
        cx<T>* Yb = X + (size_t)n * (C + 1);
        cx<T>* tw2 = Yb + (size_t)n * (C + 1);
        gen_stage_tw<T>(tw2, prm.tw, n, ctx);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int pt = p0 + k * STEP;
            k1[k] = {(T)0, (T)0};
            k2[k] = {(T)0, (T)0};
            if (live && pt < n) { k1[k] = A1[(size_t)pt * ps]; k2[k] = A2[(size_t)pt * ps]; }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int pt = p0 + k * STEP;
            if (pt < n) X[pt * (C + 1) + cl] = k1[k];
        }
        ctx.sync();
        cx<T>* src = X;
        cx<T>* dst = Yb;
        gen_line_axis<T, C>(src, dst, tw2, prm.radix, prm.n_radix, n, +1, ctx);
        // own points of the transformed A1 into registers, A2' into the OTHER buffer (nobody reads it now), then transform it
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int pt = p0 + k * STEP;
            if (pt < n) { k1[k] = src[pt * (C + 1) + cl]; dst[pt * (C + 1) + cl] = k2[k]; }
        }
        ctx.sync();
        { cx<T>* t = src; src = dst; dst = t; }
        gen_line_axis<T, C>(src, dst, tw2, prm.radix, prm.n_radix, n, +1, ctx);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int pt = p0 + k * STEP;
            if (pt < n) src[pt * (C + 1) + cl] = cmul(k1[k], src[pt * (C + 1) + cl]);
        }
        ctx.sync();
        gen_line_axis<T, C>(src, dst, tw2, prm.radix, prm.n_radix, n, -1, ctx);
        if (live) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int pt = p0 + k * STEP;
                if (pt < n) A1[(size_t)pt * ps] = src[pt * (C + 1) + cl];
            }
        }
        return;
    }
    for (int pc = p0; pc < n; pc += CH * STEP) {
        cx<T> v1[CH], v2[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int pt = pc + i * STEP;
            v1[i] = {(T)0, (T)0};
            v2[i] = {(T)0, (T)0};
            if (live && pt < n) { v1[i] = A1[(size_t)pt * ps]; v2[i] = A2[(size_t)pt * ps]; }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int pt = pc + i * STEP;
            if (pt < n) { X[pt * (C + 1) + cl] = v1[i]; Z[pt * (C + 1) + cl] = v2[i]; }
        }
    }
    ctx.sync();
    cx<T>* src = X;
    cx<T>* dst = Y;
    gen_line_axis<T, C>(src, dst, twl, prm.radix, prm.n_radix, n, +1, ctx);
    cx<T>* r1 = src;                         // A1 along x; dst is free
    src = Z;
    gen_line_axis<T, C>(src, dst, twl, prm.radix, prm.n_radix, n, +1, ctx);
    for (int pt = p0; pt < n; pt += STEP) r1[pt * (C + 1) + cl] = cmul(r1[pt * (C + 1) + cl], src[pt * (C + 1) + cl]);
    ctx.sync();
    dst = src;                               // A2 is consumed: its buffer is the forward transform's second one
    src = r1;
    gen_line_axis<T, C>(src, dst, twl, prm.radix, prm.n_radix, n, -1, ctx);
    if (live)
        for (int pt = p0; pt < n; pt += STEP) A1[(size_t)pt * ps] = src[pt * (C + 1) + cl];
}

// KA of the cubes in size-generic form: the phase multiply (compute_alpha_times_f_hat, Kernels.cu:21-59) and the (y,z) part
// of BOTH inverse transforms of one direction's x-plane, straight from f_hat.  grid = (nx planes, members x directions).
// f_hat and the phase factor of every element the thread owns are formed once and kept in registers; sign 0 takes f_hat ph,
// sign 1 f_hat conj(ph) (alpha2 = conj(alpha1), FFTWBoltzmannOperator.cpp:219-224).  out: member m at + m out_mstride,
// direction d's two planes at batch slots 2d, 2d + 1.
template <typename T>
constexpr int gen_plane_maxe() { return (int)((size_t)40 * 1024 / (2 * sizeof(cx<T>)) + GEN_THREADS - 1) / GEN_THREADS; }
template <typename T, class Ctx>
BFSM_HD void body_gen_plane_pair(const GenFftParams<T>& prm, Ctx& ctx) {
    constexpr int MAXE = gen_plane_maxe<T>();
    const int nx = prm.nx, ny = prm.ny, nz = prm.nz;
    const int ix = ctx.bx();
    const int mem = ctx.by() / prm.mper, d = ctx.by() - mem * prm.mper;
    const int LSZ = nz + 1;
    cx<T>* buf0 = ctx.template lds<cx<T>>();
    cx<T>* buf1 = buf0 + (size_t)ny * LSZ;
    cx<T>* twa = buf1 + (size_t)ny * LSZ;
    cx<T>* twb = twa + ny;
    gen_stage_tw<T>(twa, prm.tw, ny, ctx);
    gen_stage_tw<T>(twb, prm.tw_b, nz, ctx);
    const int plane = ny * nz;
    const size_t G = (size_t)nx * plane, base = (size_t)ix * plane;
    const int dy = GEN_THREADS / nz, dz = GEN_THREADS - dy * nz;
    const int iy0 = ctx.tid() / nz, iz0 = ctx.tid() - iy0 * nz;
    const cx<T>* fh = static_cast<const cx<T>*>(prm.in) + (size_t)mem * prm.in_mstride + base;
    const size_t gd = (size_t)(prm.dir0 + d);
    const cx<T> px = prm.phx[gd * nx + ix];
    cx<T> fr[MAXE], ph[MAXE];
    {
        int iy = iy0, iz = iz0;
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
            const int e = ctx.tid() + k * GEN_THREADS;
            fr[k] = {(T)0, (T)0};
            ph[k] = {(T)0, (T)0};
            if (e < plane) {
                fr[k] = fh[e];
                ph[k] = cmul(px, cmul(prm.phy[gd * ny + iy], prm.phz[gd * nz + iz]));
            }
            iy += dy; iz += dz;
            if (iz >= nz) { iz -= nz; ++iy; }
        }
    }
    for (int sgn2 = 0; sgn2 < 2; ++sgn2) {
        int iy = iy0, iz = iz0;
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
            if (ctx.tid() + k * GEN_THREADS < plane) buf0[iy * LSZ + iz] = sgn2 ? cmulc(fr[k], ph[k]) : cmul(fr[k], ph[k]);
            iy += dy; iz += dz;
            if (iz >= nz) { iz -= nz; ++iy; }
        }
        ctx.sync();
        cx<T>* src = buf0;
        cx<T>* dst = buf1;
        gen_plane_axis<T>(src, dst, twa, prm.radix, prm.n_radix, ny, +1, nz, LSZ, 1, ctx);
        gen_plane_axis<T>(src, dst, twb, prm.radix_b, prm.n_radix_b, nz, +1, ny, 1, LSZ, ctx);
        // a thread stores its own elements of src; what it then overwrites in buf0 for the second sign it has read itself (src =
        // buf0), or nobody reads any more (src = buf1: the last pass' reads of buf0 ended before its barrier)
        cx<T>* out = prm.out + (size_t)mem * prm.out_mstride + (size_t)(2 * d + sgn2) * G + base;
        iy = iy0; iz = iz0;
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
            const int e = ctx.tid() + k * GEN_THREADS;
            if (e < plane) out[e] = src[iy * LSZ + iz];
            iy += dy; iz += dz;
            if (iz >= nz) { iz -= nz; ++iy; }
        }
    }
}

// (y,z) forward transform + accumulate, KC of the cubes in size-generic form.  grid = (nx planes, groups of directions).
// The workgroup streams the P' planes of its directions once, summing dirw_d P'_d in LDS while the radial node stays the
// same (the transform is linear: one transform per run of equal radial nodes, like kc_sum_before_transform on the
// cubes), transforms the sum, weights it with beta1[r](|l|^2) (atomic_tensor_contraction, Kernels.cu:79-123, without the
// atomics) and adds it to a spectral accumulator; one slab plane per workgroup leaves at the end.  LDS: three planes.
// grid.y = members x groups for a batch of distributions.
template <typename T, class Ctx>
BFSM_HD void body_gen_plane_acc(const GenPlaneAccParams<T>& prm, Ctx& ctx) {
    const int nx = prm.nx, ny = prm.ny, nz = prm.nz;
    const int ix = ctx.bx(), mem = ctx.by() / prm.groups, g = ctx.by() - mem * prm.groups;
    const int LSZ = nz + 1;
    cx<T>* S = ctx.template lds<cx<T>>();
    cx<T>* W = S + (size_t)ny * LSZ;
    cx<T>* Qa = W + (size_t)ny * LSZ;
    cx<T>* twa = Qa + (size_t)ny * LSZ;
    cx<T>* twb = twa + nz;
    gen_stage_tw<T>(twa, prm.tw, nz, ctx);
    gen_stage_tw<T>(twb, prm.tw_b, ny, ctx);
    const size_t plane = (size_t)ny * nz, base = (size_t)ix * plane;
    const int dy = GEN_THREADS / nz, dz = GEN_THREADS - dy * nz;
    const int iy0 = ctx.tid() / nz, iz0 = ctx.tid() - iy0 * nz;
    int iy = iy0, iz = iz0;
    for (size_t e = (size_t)ctx.tid(); e < plane; e += GEN_THREADS) {
        S[iy * LSZ + iz] = {(T)0, (T)0};
        Qa[iy * LSZ + iz] = {(T)0, (T)0};
        iy += dy; iz += dz;
        if (iz >= nz) { iz -= nz; ++iy; }
    }
    const int mx = gen_mode(ix, nx);
    // transform the sum S of the finished run, weight it, add it to Qa, clear S.  Every thread touches its own elements
    // outside the transform, whose passes end in barriers.
    auto flush = [&](int r) {
        ctx.sync();
        cx<T>* src = S;
        cx<T>* dst = W;
        gen_plane_axis<T>(src, dst, twa, prm.radix, prm.n_radix, nz, -1, ny, 1, LSZ, ctx);
        gen_plane_axis<T>(src, dst, twb, prm.radix_b, prm.n_radix_b, ny, -1, nz, LSZ, 1, ctx);
        const T* b1 = prm.beta1 + (size_t)r * prm.n2stride;
        int jy = iy0, jz = iz0;
        for (size_t e = (size_t)ctx.tid(); e < plane; e += GEN_THREADS) {
            const int my = gen_mode(jy, ny), mz = gen_mode(jz, nz);
            const T b = b1[mx * mx + my * my + mz * mz];
            const cx<T> v = src[jy * LSZ + jz];
            cx<T> q = Qa[jy * LSZ + jz];
            q.x += b * v.x;
            q.y += b * v.y;
            Qa[jy * LSZ + jz] = q;
            S[jy * LSZ + jz] = {(T)0, (T)0};
            jy += dy; jz += dz;
            if (jz >= nz) { jz -= nz; ++jy; }
        }
        ctx.sync();                          // src may be W: the next run's transform overwrites it
    };
    const cx<T>* pm = prm.p + (size_t)mem * prm.p_mstride;
    const int d0 = g * prm.per_group;
    int d1 = d0 + prm.per_group;
    if (d1 > prm.n) d1 = prm.n;
    int rcur = d0 < d1 ? prm.rdir[prm.dir0 + d0] : 0;
    constexpr int CH = 4;                    // directions whose loads are in flight together
    int d = d0;
    while (d < d1) {
        const int r = prm.rdir[prm.dir0 + d];
        if (r != rcur) { flush(rcur); rcur = r; }
        int m = 1;
        while (m < CH && d + m < d1 && prm.rdir[prm.dir0 + d + m] == r) ++m;
        T w[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) w[i] = i < m ? prm.dirw[prm.dir0 + d + i] : (T)0;
        int jy = iy0, jz = iz0;
        for (size_t e = (size_t)ctx.tid(); e < plane; e += GEN_THREADS) {
            cx<T> v[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                v[i] = {(T)0, (T)0};
                if (i < m) v[i] = pm[(size_t)(d + i) * prm.p_bstride + base + e];
            }
            cx<T> s = S[jy * LSZ + jz];
#pragma unroll
            for (int i = 0; i < CH; ++i) { s.x += w[i] * v[i].x; s.y += w[i] * v[i].y; }
            S[jy * LSZ + jz] = s;
            jy += dy; jz += dz;
            if (jz >= nz) { jz -= nz; ++jy; }
        }
        d += m;
    }
    if (d0 < d1) flush(rcur);
    cx<T>* out = prm.slab + ((size_t)mem * prm.groups + g) * ((size_t)nx * plane) + base;
    iy = iy0; iz = iz0;
    for (size_t e = (size_t)ctx.tid(); e < plane; e += GEN_THREADS) {
        out[e] = Qa[iy * LSZ + iz];
        iy += dy; iz += dz;
        if (iz >= nz) { iz -= nz; ++iy; }
    }
}

template <typename T, class Ctx>
BFSM_HD void body_gen_acc(const GenAccParams<T>& prm, Ctx& ctx) {
    const size_t G = (size_t)prm.nx * prm.ny * prm.nz;
    const size_t idx = (size_t)ctx.bx() * ctx.nthreads() + ctx.tid();
    if (idx >= G) return;
    const int iz = (int)(idx % prm.nz), iy = (int)((idx / prm.nz) % prm.ny), ix = (int)(idx / ((size_t)prm.nz * prm.ny));
    const int mx = gen_mode(ix, prm.nx), my = gen_mode(iy, prm.ny), mz = gen_mode(iz, prm.nz);
    const int n2 = mx * mx + my * my + mz * mz;
    const cx<T>* pp = prm.p + (size_t)ctx.by() * prm.p_mstride;
    cx<T>* qh = prm.qhat + (size_t)ctx.by() * prm.q_mstride;
    cx<T> q = prm.first ? cx<T>{(T)0, (T)0} : qh[idx];
    if (!prm.dirw) {                          // no weights: the members are slabs of the fused sequence, already weighted
        int d = 0;
        for (; d + 8 <= prm.n; d += 8) {          // eight loads in flight, added in the same fixed order
            cx<T> t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = pp[(size_t)(d + i) * prm.p_bstride + idx];
#pragma unroll
            for (int i = 0; i < 8; ++i) { q.x += t[i].x; q.y += t[i].y; }
        }
        for (; d < prm.n; ++d) {
            const cx<T> t = pp[(size_t)d * prm.p_bstride + idx];
            q.x += t.x;
            q.y += t.y;
        }
        qh[idx] = q;
        return;
    }
    for (int d = 0; d < prm.n; ++d) {
        const size_t b = (size_t)(prm.dir0 + d);
        const T w = prm.dirw[b] * prm.beta1[(size_t)prm.rdir[b] * prm.n2stride + n2];
        const cx<T> t = pp[(size_t)d * prm.p_bstride + idx];
        q.x += w * t.x;
        q.y += w * t.y;
    }
    qh[idx] = q;
}

template <typename T, class Ctx>
BFSM_HD void body_gen_combine(const GenCombineParams<T>& prm, Ctx& ctx) {
    const size_t idx = (size_t)ctx.bx() * ctx.nthreads() + ctx.tid();
    if (idx >= prm.G) return;
    double q = (double)prm.g[idx].x;
    if (prm.with_loss) q -= (double)prm.l[idx].x * prm.f[idx];
    prm.Q[idx] = q;
}

enum class GK { Fft, Acc, Combine, FftBig, Plane, Line3, PlaneAcc, PlanePair, Fft8, FftBig8, Line38 };   // ...8: 8 lines per workgroup   // FftBig: Fft + the radix-7 / 11 / 13 passes; Plane: y and z
                                                                      // pass fused; Line3 / PlaneAcc: the fused sequence

inline bool gen_factor(int n, std::vector<int>& radix) {
    radix.clear();
    if (n < 2 || n > GEN_MAX_N) return false;
    int k2 = 0;                                    // the power of two goes in the fewest passes of radix 8, 4, 2
    while (n % 2 == 0) { ++k2; n /= 2; }
    for (int passes = (k2 + 2) / 3; passes > 0; --passes) {
        const int bits = (k2 + passes - 1) / passes;
        radix.push_back(1 << bits);
        k2 -= bits;
    }
    while (n % 3 == 0) { radix.push_back(3); n /= 3; }
    while (n % 5 == 0) { radix.push_back(5); n /= 5; }
    for (int r : {7, 11, 13}) while (n % r == 0) { radix.push_back(r); n /= r; }
    return n == 1 && radix.size() <= 8;
}

// radices served by the table-driven butterflies (GK::FftBig only)
inline bool gen_table_radix(int r) { return r == 7 || r == 11 || r == 13; }

// grids the generic path serves
inline bool gen_supported(int nx, int ny, int nz) {
    std::vector<int> r;
    for (int n : {nx, ny, nz})
        if (n < 4 || n % 2 != 0 || !gen_factor(n, r)) return false;
    return true;
}

// Backend additionally supplies:
//   template <GK kind, typename T, class P> void launch_gen(int grid_x, int grid_y, int threads, size_t lds_bytes, const P&)
template <typename T, class Backend>
struct GenericPipeline {
    PlanInfo plan;            // N = 0 marks a generic plan; directions, shard and chunks as in the fused pipeline
    Backend* be = nullptr;
    int nx = 0, ny = 0, nz = 0;
    size_t G = 0;
    int n_gl = 0;
    int chunk = 1;            // directions resident at once
    int max_batch = 1;
    int n2stride = 0;
    cx<T>* fhat = nullptr;    // [G]
    cx<T>* qhat = nullptr;    // [G]
    cx<T>* tail = nullptr;    // [2][G]: gain, loss
    cx<T>* a = nullptr;       // [2 * chunk][G]: A1, A2 interleaved per direction; P in the even members
    cx<T>* slab = nullptr;    // fused sequence: [groups][G] partial sums of one chunk
    int slab_groups = 0;
    cx<T>* tw[3] = {nullptr, nullptr, nullptr};
    cx<T>* phx = nullptr;
    cx<T>* phy = nullptr;
    cx<T>* phz = nullptr;
    T* dirw = nullptr;
    int* rdir = nullptr;
    T* beta1 = nullptr;
    T* beta2 = nullptr;
    std::vector<int> radix[3];

    template <typename U>
    bool dev_copy(U*& dst, const std::vector<U>& src) {
        dst = (U*)be->alloc((src.empty() ? 1 : src.size()) * sizeof(U));
        if (!dst) return false;
        if (!src.empty()) be->upload(dst, src.data(), src.size() * sizeof(U));
        return true;
    }

    int init(const bfsm_desc& d, Backend* backend, std::string& err) {
        be = backend;
        nx = d.nvx; ny = d.nvy; nz = d.nvz;
        G = (size_t)nx * ny * nz;
        n_gl = d.n_gl;
        max_batch = d.max_batch > 1 ? d.max_batch : 1;
        if (!gen_factor(nx, radix[0]) || !gen_factor(ny, radix[1]) || !gen_factor(nz, radix[2])) {
            err = "grid size without a kernel";
            return BFSM_ERR_UNSUPPORTED;
        }
        const long long B = (long long)d.n_gl * d.n_sph;
        plan = PlanInfo();
        plan.gen_plane = plane_ok();
        plan.gen_fused = fused_ok();
        plan.gen_moves = fused_ok() ? 6 : (line3_ok((size_t)150 * 1024) ? (plane_ok() ? 8 : 14) : (plane_ok() ? 12 : 18));                // exact_reductions / hermitian / antipodal stay false: this path evaluates every
        plan.N = 0;                       // direction whatever the flags say (include/bfsm.h documents them as no-ops here)
        plan.Gtot = G;
        plan.precision = d.precision;
        plan.n_gl = d.n_gl; plan.n_sph = d.n_sph; plan.sph_eff = d.n_sph;
        if (d.dir_begin == 0 && d.dir_end == 0) { plan.full_begin = 0; plan.full_end = B; }
        else { plan.full_begin = d.dir_begin; plan.full_end = d.dir_end; }
        plan.dir_begin = plan.full_begin; plan.dir_end = plan.full_end;
        const long long nd = plan.n_dirs();
        // directions resident at once: bounded by max_chunk (default 256) and by 8 GiB of A1 / A2 scratch
        long long c = d.max_chunk > 0 ? d.max_chunk : 256;
        const bool together = max_batch > 1 && batch_together();      // every member has its own scratch then
        const int mb = together ? max_batch : 1;
        const long long by_mem = (long long)((8.0 * 1024 * 1024 * 1024) / (2.0 * (double)G * sizeof(cx<T>) * mb));
        if (c > by_mem) c = by_mem;
        if (c > 32767 / mb) c = 32767 / mb;   // members x 2 x chunk is a grid dimension
        if (c > nd) c = nd;
        if (c < 1) c = 1;
        chunk = (int)c;
        plan.max_chunk = chunk;
        plan.largest_chunk = nd > 0 ? chunk : 0;
        for (long long o = 0; o < nd; o += chunk) {
            Chunk ck{};
            ck.dir0 = o;
            ck.n = (int)(o + chunk <= nd ? chunk : nd - o);
            plan.chunks.push_back(ck);
        }
        const int hx = nx - nx / 2, hy = ny - ny / 2, hz = nz - nz / 2;   // largest |mode| per axis
        n2stride = hx * hx + hy * hy + hz * hz + 1;
        plan.n2stride = n2stride;
        // tables (host, long double trigonometry like the fused pipeline)
        const double pi = 3.14159265358979323846;
        const long double PI_L = 3.141592653589793238462643383279502884L;
        const double fft_scale = 1.0 / (double)G;
        bool ok = true;
        const int nn[3] = {nx, ny, nz};
        for (int ax = 0; ax < 3; ++ax) {
            std::vector<cx<T>> t(nn[ax]);
            for (int k = 0; k < nn[ax]; ++k) {
                const long double ang = -2.0L * PI_L * (long double)k / (long double)nn[ax];
                t[k] = {(T)cosl(ang), (T)sinl(ang)};
            }
            ok = ok && dev_copy(tw[ax], t);
        }
        std::vector<cx<T>> hx_((size_t)nd * nx), hy_((size_t)nd * ny), hz_((size_t)nd * nz);
        std::vector<T> hw((size_t)nd);
        std::vector<int> hr((size_t)nd);
        for (long long i = 0; i < nd; ++i) {
            const long long b = plan.dir_begin + i;
            const int r = (int)(b / d.n_sph), s = (int)(b % d.n_sph);
            const long double k = -((long double)pi / (2.0L * (long double)d.L)) * (long double)d.gl_nodes[r];
            auto fill = [&](std::vector<cx<T>>& dst, int n, double sig, double scale) {
                for (int j = 0; j < n; ++j) {
                    const int l = j < n / 2 ? j : j - n;
                    const long double ang = k * (long double)l * (long double)sig;
                    dst[(size_t)i * n + j] = {(T)(scale * (double)cosl(ang)), (T)(scale * (double)sinl(ang))};
                }
            };
            fill(hx_, nx, d.sx[s], fft_scale);
            fill(hy_, ny, d.sy[s], 1.0);
            fill(hz_, nz, d.sz[s], 1.0);
            hw[(size_t)i] = (T)(fft_scale * d.gl_wts[r] * d.sph_wts[s] * std::pow(d.gl_nodes[r], d.gamma + 2));
            hr[(size_t)i] = r;
        }
        std::vector<T> b1((size_t)d.n_gl * n2stride), b2v((size_t)n2stride);
        for (int n2 = 0; n2 < n2stride; ++n2) {
            const double norm_l = std::sqrt((double)n2);
            double acc = 0;
            for (int r = 0; r < d.n_gl; ++r) {
                b1[(size_t)r * n2stride + n2] = (T)(4 * pi * d.b_gamma * sincc_ref(pi * d.gl_nodes[r] * norm_l / (2 * d.L)));
                acc += 16 * pi * pi * d.b_gamma * d.gl_wts[r] * std::pow(d.gl_nodes[r], d.gamma + 2) *
                       sincc_ref(pi * d.gl_nodes[r] * norm_l / d.L);
            }
            b2v[n2] = (T)(fft_scale * acc);
        }
        ok = ok && dev_copy(phx, hx_) && dev_copy(phy, hy_) && dev_copy(phz, hz_) && dev_copy(dirw, hw) && dev_copy(rdir, hr);
        ok = ok && dev_copy(beta1, b1) && dev_copy(beta2, b2v);
        ok = ok && (fhat = (cx<T>*)be->alloc((size_t)mb * G * sizeof(cx<T>)));
        ok = ok && (qhat = (cx<T>*)be->alloc((size_t)mb * G * sizeof(cx<T>)));
        ok = ok && (tail = (cx<T>*)be->alloc((size_t)2 * max_batch * G * sizeof(cx<T>)));
        ok = ok && (a = (cx<T>*)be->alloc((size_t)mb * 2 * chunk * G * sizeof(cx<T>)));
        if (fused_ok()) {
            slab_groups = groups_for(chunk);
            ok = ok && (slab = (cx<T>*)be->alloc((size_t)mb * slab_groups * G * sizeof(cx<T>)));
            for (const Chunk& ck : plan.chunks) plan.gen_slabs += groups_for(ck.n);
        }
        if (!ok) { err = "device allocation failed"; return BFSM_ERR_NOMEM; }
        return BFSM_OK;
    }

    void destroy() {
        if (!be) return;
        void* ptrs[] = {fhat, qhat, tail, a, slab, tw[0], tw[1], tw[2], phx, phy, phz, dirw, rdir, beta1, beta2};
        for (void* p : ptrs) if (p) be->release(p);
        fhat = qhat = tail = a = slab = phx = phy = phz = nullptr;
        tw[0] = tw[1] = tw[2] = nullptr;
        dirw = beta1 = beta2 = nullptr;
        rdir = nullptr;
    }

    int axis_len(int ax) const { return ax == 0 ? nx : (ax == 1 ? ny : nz); }
    bool fuse_reduce() const { return false; }
    bool small_path(int) const { return false; }
    void collide_small(double*, const double*, bool) {}

    // one axis pass over `batch` arrays; member b of in / out sits at + b * in_bstride / + b * out_bstride.  A pass may
    // run in place: every workgroup reads its whole block of lines into LDS before it stores the same block.
    void pass(const void* in, const cx<T>* in2, cx<T>* out, int batch, int axis, int sign, int mode, size_t in_bstride,
              size_t out_bstride, long long dir0 = 0, int mper = 0, size_t in_mstride = 0, size_t out_mstride = 0) {
        GenFftParams<T> p{};
        p.mper = mper; p.in_mstride = in_mstride; p.out_mstride = out_mstride;
        p.in = in; p.in2 = in2; p.out = out; p.tw = tw[axis];
        p.nx = nx; p.ny = ny; p.nz = nz; p.axis = axis; p.sign = sign;
        const int n = axis_len(axis);
        const int ncols = (int)(G / (size_t)n);
        // 16 lines per workgroup, 8 where 16 would leave one workgroup per CU (two line buffers above 80 KB: n > 147 in
        // double precision) -- 128-byte instead of 256-byte runs, but three workgroups per CU instead of one
        const int C = pass_lines(n);
        p.C = C;
        p.n_radix = (int)radix[axis].size();
        for (int i = 0; i < p.n_radix; ++i) p.radix[i] = radix[axis][i];
        p.mode = mode; p.phx = phx; p.phy = phy; p.phz = phz; p.dir0 = dir0; p.beta2 = beta2;
        p.in_bstride = in_bstride; p.out_bstride = out_bstride;
        const size_t lds = ((size_t)2 * n * (C + 1) + n) * sizeof(cx<T>);
        bool big = false;
        for (int r : radix[axis]) big = big || gen_table_radix(r);
        if (big && C == 8) be->template launch_gen<GK::FftBig8, T>((ncols + C - 1) / C, batch, GEN_THREADS, lds, p);
        else if (big) be->template launch_gen<GK::FftBig, T>((ncols + C - 1) / C, batch, GEN_THREADS, lds, p);
        else if (C == 8) be->template launch_gen<GK::Fft8, T>((ncols + C - 1) / C, batch, GEN_THREADS, lds, p);
        else be->template launch_gen<GK::Fft, T>((ncols + C - 1) / C, batch, GEN_THREADS, lds, p);
    }
    int pass_lines(int n) const {
#ifdef BFSM_GEN_LINES16           // A/B builds (tools only)
        return GEN_C;
#else
        return ((size_t)2 * n * (GEN_C + 1) + n) * sizeof(cx<T>) > (size_t)80 * 1024 ? 8 : GEN_C;
#endif
    }
    // ... and the x-line kernel with its three buffers
    int line3_lines() const {
#ifdef BFSM_GEN_LINES16
        return GEN_C;
#else
        return ((size_t)3 * nx * (GEN_C + 1) + nx) * sizeof(cx<T>) > (size_t)80 * 1024 ? 8 : GEN_C;
#endif
    }

    // the fused (y,z) plane kernel serves boxes whose plane fits the LDS twice and whose y / z factors are 2, 3, 5
    bool plane_ok() const {
#ifdef BFSM_GEN_NO_PLANE          // A/B builds (tools only): one pass per axis everywhere
        return false;
#else
        for (int ax = 1; ax <= 2; ++ax) for (int r : radix[ax]) if (gen_table_radix(r)) return false;
        // ... and leaves room for four workgroups per CU: with bigger planes the one-pass-per-axis kernels (52 KB at most) win
        // (64 x 48 x 80 in double precision, a 124 KB plane: 0.92 against 1.30 TB/s algorithmic, profiles/r04_generic_ktimes.txt)
        return (size_t)2 * ny * (nz + 1) * sizeof(cx<T>) <= (size_t)40 * 1024 && nz <= GEN_THREADS;
#endif
    }
    void plane(const void* in, const cx<T>* in2, cx<T>* out, int batch, int sign, int mode, size_t in_bstride, size_t out_bstride,
               long long dir0 = 0, int mper = 0, size_t in_mstride = 0, size_t out_mstride = 0) {
        GenFftParams<T> p{};
        p.mper = mper; p.in_mstride = in_mstride; p.out_mstride = out_mstride;
        p.in = in; p.in2 = in2; p.out = out;
        p.nx = nx; p.ny = ny; p.nz = nz; p.axis = 1; p.sign = sign; p.C = GEN_C;
        const int a = sign < 0 ? 2 : 1, bsec = sign < 0 ? 1 : 2;          // first / second axis transformed
        p.tw = tw[a]; p.tw_b = tw[bsec];
        p.n_radix = (int)radix[a].size();
        for (int i = 0; i < p.n_radix; ++i) p.radix[i] = radix[a][i];
        p.n_radix_b = (int)radix[bsec].size();
        for (int i = 0; i < p.n_radix_b; ++i) p.radix_b[i] = radix[bsec][i];
        p.mode = mode; p.phx = phx; p.phy = phy; p.phz = phz; p.dir0 = dir0; p.beta2 = beta2;
        p.in_bstride = in_bstride; p.out_bstride = out_bstride;
        const size_t lds = ((size_t)2 * ny * (nz + 1) + ny + nz) * sizeof(cx<T>);
        be->template launch_gen<GK::Plane, T>(nx, batch, GEN_THREADS, lds, p);
    }

    // The fused sequence (plane kernel straight from f_hat, x-line kernel, plane-accumulate kernel: 6 array moves per
    // direction) serves the boxes of plane_ok() whose x factors are 2, 3, 5 and whose three x-line buffers leave room for
    // two workgroups per CU.
    bool fused_ok() const {
#ifdef BFSM_GEN_NO_FUSE           // A/B builds (tools only): the 12-move sequence
        return false;
#else
        return plane_ok() && line3_ok((size_t)80 * 1024);
#endif
    }
    // the x-line kernel alone (both inverse x passes, the product and the forward x pass in one: 4 array moves fewer) also
    // serves boxes whose plane does not fit, as long as its three line buffers fit the CU
    bool line3_ok(size_t lds_cap) const {
#ifdef BFSM_GEN_NO_FUSE
        return false;
#else
        for (int r : radix[0]) if (gen_table_radix(r)) return false;
        return line3_lds() <= lds_cap;
#endif
    }
    // LDS of the x-line kernel: three line buffers, two in the 8-line form (which parks a line in registers instead)
    size_t line3_lds() const {
        const int C = line3_lines();
        return ((size_t)(C == 8 ? 2 : 3) * nx * (C + 1) + nx) * sizeof(cx<T>);
    }
    void line3(const Chunk& c, int nb = 1) {
        GenLineParams<T> kl{};
        kl.a = a; kl.tw = tw[0]; kl.nx = nx; kl.ny = ny; kl.nz = nz;
        kl.n = c.n; kl.mstride = a_mstride();
        kl.n_radix = (int)radix[0].size();
        for (int i = 0; i < kl.n_radix; ++i) kl.radix[i] = radix[0][i];
        const int ncols = ny * nz;
        be->mark(BFSM_K_GAIN_LINE, 3.0 * c.n * nb * (double)G * sizeof(cx<T>));
        const int C = line3_lines();
        const size_t lds = line3_lds();
        if (C == 8) be->template launch_gen<GK::Line38, T>((ncols + C - 1) / C, c.n * nb, GEN_THREADS, lds, kl);
        else be->template launch_gen<GK::Line3, T>((ncols + C - 1) / C, c.n * nb, GEN_THREADS, lds, kl);
    }
    // groups of directions per x-plane in the plane-accumulate kernel: about 512 workgroups per launch (two per CU; 1024 and
    // 2048 measured 3 % and 7 % slower over the evaluation at 32 x 64 x 16: more slabs to write and to sum)
    int groups_for(int n) const {
#ifdef BFSM_GEN_TARGET_WGS        // the emulator build asks for few workgroups, so that its small cases give a group several
        int g = (BFSM_GEN_TARGET_WGS + nx - 1) / nx;          // directions and runs that cross radial nodes
#else
        int g = (512 + nx - 1) / nx;
#endif
        if (g > n) g = n;
        if (g < 1) g = 1;
        const int per = (n + g - 1) / g;
        return per > 0 ? (n + per - 1) / per : 1;
    }
    // elements between the scratch of two batch members
    size_t a_mstride() const { return (size_t)2 * chunk * G; }
    // batches of distributions go through the fused sequence together (every launch covers all members); the other
    // sequences take them one after the other
    bool batch_together() const { return fused_ok(); }
    void gain_chunk_fused(const Chunk& c, bool first, int nb = 1) {
        const double Gc = (double)G * sizeof(cx<T>);
        // A1', A2' = IFFT_yz(alpha f_hat / G), IFFT_yz(conj(alpha) f_hat / G), straight from f_hat (x stays spectral)
        be->mark(BFSM_K_GAIN_INV, 2.0 * c.n * nb * Gc);
#ifdef BFSM_GEN_NO_PLANE_PAIR     // A/B builds (tools only): one workgroup per sign
        constexpr bool PAIR = false;
#else
        // both signs in one workgroup (f_hat and the phase factors formed once): double precision 0.195 -> 0.156 ms at 32 x 64 x 16;
        // single precision keeps one workgroup per sign (49 VGPRs and 18 KB of LDS: twice the residency; the pair form measured
        // 0.117 -> 0.133 ms there) -- profiles/r04_generic_fused_ab.txt
        constexpr bool PAIR = sizeof(T) == 8;
#endif
        if (!PAIR) plane(fhat, nullptr, a, 2 * c.n * nb, +1, GEN_PHASE, 0, G, c.dir0, 2 * c.n, G, a_mstride());
        else {
            GenFftParams<T> ki{};
            ki.in = fhat; ki.out = a; ki.nx = nx; ki.ny = ny; ki.nz = nz; ki.axis = 1; ki.sign = +1; ki.C = GEN_C; ki.mode = GEN_PHASE;
            ki.tw = tw[1]; ki.tw_b = tw[2];
            ki.n_radix = (int)radix[1].size(); ki.n_radix_b = (int)radix[2].size();
            for (int i = 0; i < ki.n_radix; ++i) ki.radix[i] = radix[1][i];
            for (int i = 0; i < ki.n_radix_b; ++i) ki.radix_b[i] = radix[2][i];
            ki.phx = phx; ki.phy = phy; ki.phz = phz; ki.dir0 = c.dir0;
            ki.mper = c.n; ki.in_mstride = G; ki.out_mstride = a_mstride();
            be->template launch_gen<GK::PlanePair, T>(nx, c.n * nb, GEN_THREADS, ((size_t)2 * ny * (nz + 1) + ny + nz) * sizeof(cx<T>), ki);
        }
        line3(c, nb);
        GenPlaneAccParams<T> kp{};
        const int groups = groups_for(c.n);
        kp.p = a; kp.p_bstride = 2 * G; kp.slab = slab; kp.dirw = dirw; kp.rdir = rdir; kp.beta1 = beta1;
        kp.dir0 = c.dir0; kp.n = c.n; kp.per_group = (c.n + groups - 1) / groups; kp.n2stride = n2stride;
        kp.nx = nx; kp.ny = ny; kp.nz = nz; kp.tw = tw[2]; kp.tw_b = tw[1];
        kp.n_radix = (int)radix[2].size(); kp.n_radix_b = (int)radix[1].size();
        for (int i = 0; i < kp.n_radix; ++i) kp.radix[i] = radix[2][i];
        for (int i = 0; i < kp.n_radix_b; ++i) kp.radix_b[i] = radix[1][i];
        kp.groups = groups; kp.p_mstride = a_mstride();
        be->mark(BFSM_K_GAIN_FWD, (1.0 * c.n + groups) * nb * Gc);
        be->template launch_gen<GK::PlaneAcc, T>(nx, groups * nb, GEN_THREADS, ((size_t)3 * ny * (nz + 1) + ny + nz) * sizeof(cx<T>), kp);
        // Q_hat (+)= the groups' slabs, fixed order
        GenAccParams<T> ka{slab, G, qhat, nullptr, nullptr, nullptr, 0, groups, n2stride, first ? 1 : 0, nx, ny, nz, (size_t)groups * G, G};
        be->mark(BFSM_K_REDUCE, (groups + (first ? 1.0 : 2.0)) * nb * Gc);
        be->template launch_gen<GK::Acc, T>((int)((G + GEN_THREADS - 1) / GEN_THREADS), nb, GEN_THREADS, 0, ka);
    }

    // f_hat = FFT(f), then the gain term of this shard into qhat   (CUDABoltzmannOperator.cu:131-191)
    // nb > 1 (members at f_dev + m G, their f_hat / Q_hat at fhat / qhat + m G) only where batch_together()
    void gain_partial(const double* f_dev, int nb = 1, bool = true) {
        const double Gc = (double)G * sizeof(cx<T>);
        const bool pl = plane_ok();
        be->mark(BFSM_K_FFT_F, 1.5 * nb * Gc);
        if (pl) plane(f_dev, nullptr, fhat, nb, -1, GEN_REAL, G, G);
        else {
            pass(f_dev, nullptr, fhat, nb, 2, -1, GEN_REAL, G, G);
            be->mark(BFSM_K_FFT_F, 2.0 * nb * Gc); pass(fhat, nullptr, fhat, nb, 1, -1, GEN_PLAIN, G, G);
        }
        be->mark(BFSM_K_FFT_F, 2.0 * nb * Gc); pass(fhat, nullptr, fhat, nb, 0, -1, GEN_PLAIN, G, G);
        bool first = true;
        const bool fused = fused_ok(), l3 = line3_ok((size_t)150 * 1024);
        for (const Chunk& c : plan.chunks) {
            if (fused) { gain_chunk_fused(c, first, nb); first = false; continue; }
            const int nb2 = 2 * c.n;
            if (l3) {
                // the (y,z) passes of the inverse transforms first (phase factors on the load side, straight from f_hat), then
                // the x-line kernel, then the (y,z) passes of the forward transform: 14 array moves (8 with the plane kernel)
                be->mark(BFSM_K_GAIN_INV, 2.0 * c.n * Gc);
                if (pl) plane(fhat, nullptr, a, nb2, +1, GEN_PHASE, 0, G, c.dir0);
                else {
                    pass(fhat, nullptr, a, nb2, 2, +1, GEN_PHASE, 0, G, c.dir0);
                    be->mark(BFSM_K_GAIN_INV, 0); pass(a, nullptr, a, nb2, 1, +1, GEN_PLAIN, G, G);
                }
                line3(c);
                be->mark(BFSM_K_GAIN_LINE, 0);
                if (pl) plane(a, nullptr, a, c.n, -1, GEN_PLAIN, 2 * G, 2 * G);
                else {
                    pass(a, nullptr, a, c.n, 1, -1, GEN_PLAIN, 2 * G, 2 * G);
                    be->mark(BFSM_K_GAIN_LINE, 0); pass(a, nullptr, a, c.n, 2, -1, GEN_PLAIN, 2 * G, 2 * G);
                }
                GenAccParams<T> ka{a, 2 * G, qhat, dirw, rdir, beta1, c.dir0, c.n, n2stride, first ? 1 : 0, nx, ny, nz};
                be->mark(BFSM_K_GAIN_FWD, 1.0 * c.n * Gc);
                be->template launch_gen<GK::Acc, T>((int)((G + GEN_THREADS - 1) / GEN_THREADS), 1, GEN_THREADS, 0, ka);
                first = false;
                continue;
            }
            // A1, A2 = IFFT(alpha f_hat / G), IFFT(conj(alpha) f_hat / G): members 2d, 2d + 1 of `a`
            // accounting: the SURVEY 8(d) model attributes 2 array passes per direction to this group (the inverse
            // transforms), 3 to the next (product + forward transform) and 1 to the accumulate; the path moves more
            be->mark(BFSM_K_GAIN_INV, 2.0 * c.n * Gc);
            pass(fhat, nullptr, a, nb2, 0, +1, GEN_PHASE, 0, G, c.dir0);
            if (pl) { be->mark(BFSM_K_GAIN_INV, 0); plane(a, nullptr, a, nb2, +1, GEN_PLAIN, G, G); }
            else {
                be->mark(BFSM_K_GAIN_INV, 0); pass(a, nullptr, a, nb2, 1, +1, GEN_PLAIN, G, G);
                be->mark(BFSM_K_GAIN_INV, 0); pass(a, nullptr, a, nb2, 2, +1, GEN_PLAIN, G, G);
            }
            // P_hat = FFT(A1 * A2): the product is formed on the load side of the first forward pass and written over
            // A1 (members 2d, stride 2G), then transformed in place
            be->mark(BFSM_K_GAIN_LINE, 3.0 * c.n * Gc);
            if (pl) plane(a, a + G, a, c.n, -1, GEN_PRODUCT, 2 * G, 2 * G);
            else {
                pass(a, a + G, a, c.n, 2, -1, GEN_PRODUCT, 2 * G, 2 * G);
                be->mark(BFSM_K_GAIN_LINE, 0); pass(a, nullptr, a, c.n, 1, -1, GEN_PLAIN, 2 * G, 2 * G);
            }
            be->mark(BFSM_K_GAIN_LINE, 0); pass(a, nullptr, a, c.n, 0, -1, GEN_PLAIN, 2 * G, 2 * G);
            GenAccParams<T> ka{a, 2 * G, qhat, dirw, rdir, beta1, c.dir0, c.n, n2stride, first ? 1 : 0, nx, ny, nz};
            be->mark(BFSM_K_GAIN_FWD, 1.0 * c.n * Gc);
            be->template launch_gen<GK::Acc, T>((int)((G + GEN_THREADS - 1) / GEN_THREADS), 1, GEN_THREADS, 0, ka);
            first = false;
        }
        if (first) {   // empty shard: qhat = 0
            GenAccParams<T> ka{a, 2 * G, qhat, dirw, rdir, beta1, 0, 0, n2stride, 1, nx, ny, nz, 0, G};
            be->mark(-1, 0);
            be->template launch_gen<GK::Acc, T>((int)((G + GEN_THREADS - 1) / GEN_THREADS), nb, GEN_THREADS, 0, ka);
        }
    }

    // loss term + final inverse transforms + combine   (CUDABoltzmannOperator.cu:193-216)
    void finish(double* Q_dev, const double* f_dev, bool with_loss = true, int nb = 1, bool = false) {
        const double Gc = (double)G * sizeof(cx<T>);
        cx<T>* tg = tail;                                   // [max_batch][G] gain, then [max_batch][G] loss
        cx<T>* tl = tail + (size_t)max_batch * G;
        be->mark(BFSM_K_TAIL, (with_loss ? 7.0 : 3.5) * nb * Gc);
        const bool pl = plane_ok();
        auto yz = [&](cx<T>* t) {
            if (pl) { be->mark(BFSM_K_TAIL, 0); plane(t, nullptr, t, nb, +1, GEN_PLAIN, G, G); }
            else {
                be->mark(BFSM_K_TAIL, 0); pass(t, nullptr, t, nb, 1, +1, GEN_PLAIN, G, G);
                be->mark(BFSM_K_TAIL, 0); pass(t, nullptr, t, nb, 2, +1, GEN_PLAIN, G, G);
            }
        };
        if (with_loss) {
            // gain and loss term through the same launches: slot s of member m at tail + (s max_batch + m) G
            const size_t half = (size_t)max_batch * G;
            pass(qhat, fhat, tail, 2 * nb, 0, +1, GEN_TAIL2, 0, half, 0, 2, G, G);
            if (pl) { be->mark(BFSM_K_TAIL, 0); plane(tail, nullptr, tail, 2 * nb, +1, GEN_PLAIN, G, G, 0, nb, half, half); }
            else {
                be->mark(BFSM_K_TAIL, 0); pass(tail, nullptr, tail, 2 * nb, 1, +1, GEN_PLAIN, G, G, 0, nb, half, half);
                be->mark(BFSM_K_TAIL, 0); pass(tail, nullptr, tail, 2 * nb, 2, +1, GEN_PLAIN, G, G, 0, nb, half, half);
            }
        } else {
            pass(qhat, nullptr, tg, nb, 0, +1, GEN_PLAIN, G, G);
            yz(tg);
        }
        const size_t tot = (size_t)nb * G;                  // members are contiguous in every array involved
        GenCombineParams<T> kc{tg, tl, f_dev, Q_dev, tot, with_loss ? 1 : 0};
        be->mark(BFSM_K_TAIL, 0);
        be->template launch_gen<GK::Combine, T>((int)((tot + GEN_THREADS - 1) / GEN_THREADS), 1, GEN_THREADS, 0, kc);
    }

    // In-place batched 3-D transform on user data (bfsm_fft3d); natural layouts on both sides
    int fft3d(cx<T>* data, int batch, int sign) {
        if (plane_ok()) {
            if (sign < 0) { be->mark(-1, 0); plane(data, nullptr, data, batch, -1, GEN_PLAIN, G, G); }
            be->mark(-1, 0);
            pass(data, nullptr, data, batch, 0, sign, GEN_PLAIN, G, G);
            if (sign > 0) { be->mark(-1, 0); plane(data, nullptr, data, batch, +1, GEN_PLAIN, G, G); }
            return BFSM_OK;
        }
        const int order[3] = {2, 1, 0};
        for (int i = 0; i < 3; ++i) {
            be->mark(-1, 0);
            pass(data, nullptr, data, batch, sign < 0 ? order[i] : order[2 - i], sign, GEN_PLAIN, G, G);
        }
        return BFSM_OK;
    }
};

}  // namespace bfsm
