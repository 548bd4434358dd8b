// bfsm_core.hpp -- workgroup-level bodies of the gfx950 kernels of the Fourier-spectral Boltzmann collision
// operator (replaces the cuFFT plans + the 6 CUDA kernels of Collisions/CUDABoltzmannOperator.cu:119-220 and
// Collisions/BoltzmannCUDAKernels.cu:4-177 with a fused 3-kernel-per-chunk pipeline).
//
// Every body is a function template over a small execution context `Ctx` (thread id, block id, LDS base,
// workgroup barrier).  bfsm_hip.hip instantiates them with the device context inside __global__ kernels;
// tests/emu instantiates the same code with a host lock-step context so the index algebra is unit-tested on CPU.
//
// Data layout conventions (N = Nvx = Nvy = Nvz, G = N^3, complex = interleaved (re,im) of T):
//   physical space   [x][y][z]      z contiguous  (the reference's idx3 = (i*Nvy + j)*Nvz + k)
//   spectral space   [lx][lz][ly]   ly contiguous ("spectral-transposed": every 2-D plane transform transposes,
//                                   so no pass ever needs a strided global access)
//   Fourier modes in FFT order 0..N/2-1, -N/2..-1 (FFTWBoltzmannOperator.cpp:50-57)
//
// 1-D transforms: a line of N points is held by T = N/E threads, E points each, thread u owning the points
// u + T*m (m < E).  One Stockham step = radix-E butterflies in registers, twiddle, exchange through LDS,
// radix-T butterflies; the result is again distributed as u + T*m, so passes chain without re-distribution.
// Lanes always run along the axis that is NOT being transformed, so LDS exchanges are conflict-free and every
// global access is a run of N contiguous elements.
#pragma once
#include <stdint.h>

#ifndef BFSM_HD
#define BFSM_HD __device__ __forceinline__
#endif

namespace bfsm {

template <typename T>
struct alignas(2 * sizeof(T)) cx {
    T x, y;
};

template <typename T> BFSM_HD cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> BFSM_HD cx<T> csub(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> BFSM_HD cx<T> cmul(cx<T> a, cx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// a * conj(b)
template <typename T> BFSM_HD cx<T> cmulc(cx<T> a, cx<T> b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
// multiply by SGN * i
template <int SGN, typename T> BFSM_HD cx<T> rot90(cx<T> a) {
    if (SGN > 0) return {-a.y, a.x};
    return {a.y, -a.x};
}

// cos / sin of 2*pi*j/16, folded at compile time after unrolling
BFSM_HD constexpr double cos16(int j) {
    switch (j & 15) {
        case 0: return 1.0;
        case 1: return 0.92387953251128673848;
        case 2: return 0.70710678118654752440;
        case 3: return 0.38268343236508977173;
        case 4: return 0.0;
        case 5: return -0.38268343236508977173;
        case 6: return -0.70710678118654752440;
        case 7: return -0.92387953251128673848;
        case 8: return -1.0;
        case 9: return -0.92387953251128673848;
        case 10: return -0.70710678118654752440;
        case 11: return -0.38268343236508977173;
        case 12: return 0.0;
        case 13: return 0.38268343236508977173;
        case 14: return 0.70710678118654752440;
        default: return 0.92387953251128673848;
    }
}
BFSM_HD constexpr double sin16(int j) { return cos16(j + 12); }  // sin(a) = cos(a - pi/2) = cos(a + 3pi/2)

// Fused-multiply-add helper (one v_fma on the device; contraction is not left to the optimiser's discretion).
template <typename T> BFSM_HD T fmad(T a, T b, T c) { return __builtin_fma(a, b, c); }
template <> BFSM_HD float fmad<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Twiddled radix-2 butterfly  (a, b) <- (a + w b, a - w b),  w = (c, s), in SIX fused multiply-adds instead of a
// complex multiply (4) plus add / subtract (4):   p = a + w b  takes two chained FMAs per component, and the other
// output is 2 a - p (one FMA per component).  Rounding stays at the level of the plain form (one extra rounding of a
// quantity no larger than |a| + |b|).
template <typename T> BFSM_HD void bfly_tw(cx<T>& a, cx<T>& b, T c, T s) {
    const T px = fmad(-b.y, s, fmad(b.x, c, a.x));
    const T py = fmad(b.x, s, fmad(b.y, c, a.y));
    b = {fmad((T)2, a.x, -px), fmad((T)2, a.y, -py)};
    a = {px, py};
}

// In-register DFT of R points, natural order in and out, unnormalised.
// SGN = -1: forward (exp(-i...)), SGN = +1: backward -- the FFTW_FORWARD / FFTW_BACKWARD convention.
// Radix-2 decimation in time; butterflies with a non-trivial twiddle use bfly_tw (16 points: 148 operations).
template <int R, int SGN, typename T>
struct SmallDft {
    static BFSM_HD void run(cx<T>* a) {
        static_assert(R == 4 || R == 8 || R == 16, "radix");
#ifdef BFSM_KO_DFT      // knock-out builds (tools only): timing experiments, wrong results
        return;
#endif
        cx<T> e[R / 2], o[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            e[k] = a[2 * k];
            o[k] = a[2 * k + 1];
        }
        SmallDft<R / 2, SGN, T>::run(e);
        SmallDft<R / 2, SGN, T>::run(o);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            const int j = k * (16 / R);  // angle 2*pi*k/R in sixteenths of a turn
            if (j == 0) {
                a[k] = cadd(e[k], o[k]);
                a[k + R / 2] = csub(e[k], o[k]);
            } else if (j == 4) {
                const cx<T> t = rot90<SGN>(o[k]);
                a[k] = cadd(e[k], t);
                a[k + R / 2] = csub(e[k], t);
            } else {
                cx<T> x = e[k], y = o[k];
                bfly_tw(x, y, (T)cos16(j), (T)(SGN * sin16(j)));
                a[k] = x;
                a[k + R / 2] = y;
            }
        }
    }
};
template <int SGN, typename T>
struct SmallDft<2, SGN, T> {
    static BFSM_HD void run(cx<T>* a) {
        const cx<T> t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    }
};
template <int SGN, typename T>
struct SmallDft<1, SGN, T> {
    static BFSM_HD void run(cx<T>*) {}
};
// 3 points: 12 operations (t = b + c; y0 = a + t; m = a - t/2; y1,2 = m +- (+-i) (sqrt(3)/2) (b - c))
template <int SGN, typename T>
struct SmallDft<3, SGN, T> {
    static BFSM_HD void run(cx<T>* a) {
        constexpr T K = (T)(SGN * 0.86602540378443864676);   // sin(2 pi / 3), signed: w = exp(SGN i 2 pi / 3)
        const cx<T> t = cadd(a[1], a[2]), d = csub(a[1], a[2]);
        const cx<T> m = {fmad((T)-0.5, t.x, a[0].x), fmad((T)-0.5, t.y, a[0].y)};
        a[0] = cadd(a[0], t);
        a[1] = {fmad(-K, d.y, m.x), fmad(K, d.x, m.y)};      // m + i K d
        a[2] = {fmad(K, d.y, m.x), fmad(-K, d.x, m.y)};      // m - i K d
    }
};
// 5 points: the two symmetric / antisymmetric pairs (1,4), (2,3) combined with cos / sin of 2 pi / 5 and 4 pi / 5
template <int SGN, typename T>
struct SmallDft<5, SGN, T> {
    static BFSM_HD void run(cx<T>* a) {
        constexpr T C1 = (T)0.30901699437494742410, C2 = (T)-0.80901699437494742410;
        constexpr T S1 = (T)(SGN * 0.95105651629515357212), S2 = (T)(SGN * 0.58778525229247312917);
        const cx<T> a1 = cadd(a[1], a[4]), b1 = csub(a[1], a[4]), a2 = cadd(a[2], a[3]), b2 = csub(a[2], a[3]);
        const cx<T> m1 = {fmad(C2, a2.x, fmad(C1, a1.x, a[0].x)), fmad(C2, a2.y, fmad(C1, a1.y, a[0].y))};
        const cx<T> m2 = {fmad(C1, a2.x, fmad(C2, a1.x, a[0].x)), fmad(C1, a2.y, fmad(C2, a1.y, a[0].y))};
        const cx<T> r1 = {-fmad(S2, b2.y, S1 * b1.y), fmad(S2, b2.x, S1 * b1.x)};      // i (S1 b1 + S2 b2)
        const cx<T> r2 = {-fmad(-S1, b2.y, S2 * b1.y), fmad(-S1, b2.x, S2 * b1.x)};    // i (S2 b1 - S1 b2)
        a[0] = cadd(a[0], cadd(a1, a2));
        a[1] = cadd(m1, r1);
        a[4] = csub(m1, r1);
        a[2] = cadd(m2, r2);
        a[3] = csub(m2, r2);
    }
};
// Sizes N1 * N2 with coprime factors by the prime-factor (Good-Thomas) index maps: no twiddles between the two
// stages, only compile-time permutations of registers.
//   in:  n = (N2 n1 + N1 n2) mod N        out: k = (N2 (N2^-1 mod N1) k1 + N1 (N1^-1 mod N2) k2) mod N
constexpr int inv_mod(int a, int m) {
    for (int x = 1; x < m; ++x) if ((a * x) % m == 1) return x;
    return 1;
}
template <int N1, int N2, int SGN, typename T>
struct SmallDftPfa {
    static BFSM_HD void run(cx<T>* a) {
        constexpr int N = N1 * N2, C1 = N2 * inv_mod(N2 % N1, N1), C2 = N1 * inv_mod(N1 % N2, N2);
        cx<T> t[N2][N1];
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) {
#pragma unroll
            for (int n1 = 0; n1 < N1; ++n1) t[n2][n1] = a[(N2 * n1 + N1 * n2) % N];
            SmallDft<N1, SGN, T>::run(t[n2]);
        }
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) {
            cx<T> c[N2];
#pragma unroll
            for (int n2 = 0; n2 < N2; ++n2) c[n2] = t[n2][k1];
            SmallDft<N2, SGN, T>::run(c);
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) a[(C1 * k1 + C2 * k2) % N] = c[k2];
        }
    }
};
template <int SGN, typename T> struct SmallDft<6, SGN, T> { static BFSM_HD void run(cx<T>* a) { SmallDftPfa<2, 3, SGN, T>::run(a); } };
template <int SGN, typename T> struct SmallDft<12, SGN, T> { static BFSM_HD void run(cx<T>* a) { SmallDftPfa<4, 3, SGN, T>::run(a); } };
template <int SGN, typename T> struct SmallDft<24, SGN, T> { static BFSM_HD void run(cx<T>* a) { SmallDftPfa<8, 3, SGN, T>::run(a); } };
template <int SGN, typename T> struct SmallDft<20, SGN, T> { static BFSM_HD void run(cx<T>* a) { SmallDftPfa<4, 5, SGN, T>::run(a); } };

// The same DFT of R points whose INPUTS carry twiddle factors: x[j] is to be multiplied by w[j] first (w[0] = 1 when
// FIRST_ONE).  w holds forward-table values exp(-i...); the backward transform (SGN = +1) uses their conjugates.  The
// factors are folded into the leaf butterflies of the decimation-in-time tree: x_i w_i +- x_j w_j costs 6 operations
// when w_i = 1 and 4 + 6 otherwise, against 4 + 4 + 4 for multiply-then-butterfly (8 points: 36 instead of 44).
template <int R, int SGN, typename T, bool FIRST_ONE>
struct SmallDftTw {
    static BFSM_HD void run(cx<T>* a, const cx<T>* w) {
        static_assert(R == 4 || R == 8 || R == 16, "radix");
#ifdef BFSM_KO_DFT
        return;
#endif
        cx<T> e[R / 2], o[R / 2], we[R / 2], wo[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            e[k] = a[2 * k];
            o[k] = a[2 * k + 1];
            we[k] = w[2 * k];
            wo[k] = w[2 * k + 1];
        }
        SmallDftTw<R / 2, SGN, T, FIRST_ONE>::run(e, we);
        SmallDftTw<R / 2, SGN, T, false>::run(o, wo);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            const int j = k * (16 / R);
            if (j == 0) {
                a[k] = cadd(e[k], o[k]);
                a[k + R / 2] = csub(e[k], o[k]);
            } else if (j == 4) {
                const cx<T> t = rot90<SGN>(o[k]);
                a[k] = cadd(e[k], t);
                a[k + R / 2] = csub(e[k], t);
            } else {
                cx<T> x = e[k], y = o[k];
                bfly_tw(x, y, (T)cos16(j), (T)(SGN * sin16(j)));
                a[k] = x;
                a[k + R / 2] = y;
            }
        }
    }
};
template <int SGN, typename T, bool FIRST_ONE>
struct SmallDftTw<2, SGN, T, FIRST_ONE> {
    static BFSM_HD void run(cx<T>* a, const cx<T>* w) {
        cx<T> x = a[0], y = a[1];
        if (!FIRST_ONE) x = (SGN < 0) ? cmul(x, w[0]) : cmulc(x, w[0]);
        bfly_tw(x, y, w[1].x, (SGN < 0) ? w[1].y : -w[1].y);
        a[0] = x;
        a[1] = y;
    }
};

// Compile-time geometry of an N-point line: E points per thread, T threads per line.
template <int N> struct Geo;
template <> struct Geo<16>  { static constexpr int E = 4,  T = 4; };
template <> struct Geo<32>  { static constexpr int E = 8,  T = 4; };
template <> struct Geo<64>  { static constexpr int E = 8,  T = 8; };
template <> struct Geo<128> { static constexpr int E = 16, T = 8; };
// sizes with a factor 3: N = Q T^2 needs T = 4 (the radix-E step is a prime-factor 4 x 3 / 8 x 3 transform)
template <> struct Geo<48>  { static constexpr int E = 12, T = 4; };
template <> struct Geo<96>  { static constexpr int E = 24, T = 4; };
template <> struct Geo<80>  { static constexpr int E = 20, T = 4; };    // prime-factor 4 x 5 register transform
template <> struct Geo<24>  { static constexpr int E = 12, T = 2; };    // two threads per line (48-thread tiles)
template <> struct Geo<40>  { static constexpr int E = 20, T = 2; };

// columns per workgroup of the line (x-axis) kernels: a row of N, or a divisor of it that keeps runs >= 384 bytes; at
// N = 32 TWO adjacent rows of the inner N * N space (64 contiguous columns), so that a row of threads is a whole wave:
// one u per wave, twiddles through the scalar cache, scalar-base streams in 1-KiB runs -- what body_gain_inv_pair gave KA
// (round 4; BFSM_NO_LINE_PAIR_32 restores 32 columns).  (N = 128 with whole rows of 128 columns, 1024-thread workgroups,
// one per CU: KB 6.55 -> 7.5 ms on the 768-direction slice of config 5, profiles/r04_ka_interleave_ab2.txt.)
#ifdef BFSM_NO_LINE_PAIR_32
constexpr int line_npl(int n) { return n <= 64 ? n : (n % 64 == 0 ? 64 : n / 2); }
#else
constexpr int line_npl(int n) { return n == 32 ? 64 : (n <= 64 ? n : (n % 64 == 0 ? 64 : n / 2)); }
#endif

// Lanes per row of threads.  Rows that do not cover whole waves are padded to the next multiple of 64 lanes at N = 40,
// 48 and 96 (and their line-kernel column counts); a padding lane DUPLICATES the lane (ROW - N) below it -- same
// addresses, same values, everywhere -- so no access needs a guard, and every wave has ONE u: twiddles and phase factors
// become scalar loads as at N = 64 / 128 instead of per-lane vector loads (N = 96: 84 of them per tile and thread).
// Costs 25 - 37 % duplicate lanes.  Measured, 384 directions fp64, padded against unpadded: 96^3 8.02 / 8.51 ms (KB 2.93 /
// 3.53), 40^3 0.543 / 0.570, 48^3 0.818 / 0.803 (Hermitian mode 0.298 / 0.418); 80^3 4.90 / 4.45 ms: not padded.
constexpr int row_pad(int n) { return (n == 40 || n == 48) ? 64 : (n == 96 ? 128 : n); }

template <int N>
struct Wg {
    static constexpr int E = Geo<N>::E;
    static constexpr int T = Geo<N>::T;
    static constexpr int Q = E / T;           // radix-T sub-transforms per thread in the second step
    static constexpr int ROW = row_pad(N);    // lanes per tile row (>= N)
    static constexpr int THREADS = ROW * T;   // one N x N tile per workgroup
    static constexpr int LS = N + 1;          // LDS row stride (elements); odd => transposed reads conflict-free
    static constexpr int LDS_ELEMS = N * LS;
    // Line kernels (1-D passes along x) only need a set of independent columns, not a whole tile: they take NPL
    // columns per workgroup.  At N = 128 that is half a row, which halves the exchange buffer (66 KiB in fp32) and
    // lets two workgroups share a CU; the 2-D tile kernels keep N columns.
    static constexpr int NPL = line_npl(N);
    static constexpr int LROW = row_pad(NPL); // lanes per row of a line kernel (>= NPL)
    static constexpr int LINE_THREADS = LROW * T;
    // N rows for the exchanges; N + 2 so that the Hermitian line kernel can stage the stored halves (N/2 + 1 rows) of
    // both arrays side by side
    static constexpr int LINE_LDS_ELEMS = (N + 2) * (NPL + 1);
    static_assert(E * T == N && Q * T == E && (N * N) % NPL == 0 && (N % NPL == 0 || NPL % N == 0), "geometry");
};

// (column p, line share u) of this thread in a row of ROWW lanes serving W columns (see row_pad)
template <int W, int ROWW, class Ctx>
BFSM_HD void lane_coords(Ctx& ctx, int& p, int& u) {
    const int tid = ctx.tid(), pr = tid % ROWW;
    u = ctx.uniform(tid / ROWW, ROWW);
    p = (ROWW == W || pr < W) ? pr : pr - (ROWW - W);
}

// A whole N x N tile of complex T must fit the CU's 160 KiB LDS for the exchanges of the 2-D tile kernels.  The one
// geometry where it does not (N = 128 in fp64: 258 KiB) exchanges the real and the imaginary parts one after the
// other through a buffer of N x (N+1) scalars (129 KiB): twice the barriers, same data path.
template <int N, typename T>
constexpr bool split_tile() { return sizeof(cx<T>) * (size_t)Wg<N>::LDS_ELEMS > 160u * 1024u; }
template <int N, typename T>
constexpr size_t tile_lds_bytes() { return (size_t)Wg<N>::LDS_ELEMS * (split_tile<N, T>() ? sizeof(T) : sizeof(cx<T>)); }
template <int N, typename T>
constexpr size_t line_lds_bytes() { return (size_t)Wg<N>::LINE_LDS_ELEMS * sizeof(cx<T>); }

// ---- one distributed 1-D transform -------------------------------------------------------------------------
// v[m] = x[u + T*m] on entry, X[u + T*m] on exit.  lds rows are indexed by position along the line, columns
// by the lane index p.  tw[n] = exp(-2*pi*i*n/N) (forward table; conjugated for SGN = +1).
//
//   X[k1 + E k2] = sum_u W_T^(u k2) * W_N^(u k1) * ( sum_m x[u + T m] W_E^(m k1) )
//
// Step 1: every thread transforms its E points (radix E).  Exchange: thread u' receives, for its Q values
// k1 = u' + T q, the results of all T threads of the line.  Step 2: radix-T transforms over the source thread uu whose
// inputs carry the inter-step twiddles W_N^(uu k1); they are folded into the leaf butterflies (SmallDftTw), which is
// cheaper than multiplying before the exchange and lets the LDS stores issue straight after step 1.
// Twiddles<N,T>: this thread's (T-1) Q inter-step twiddles tw[uu * (u + T q)].  Normally they are loaded once per
// kernel (wave-uniform for N >= 64, so they live in SGPRs and cost no vector-memory latency inside the direction
// loops); in the split-exchange geometry (N = 128, fp64: SGPRs of twiddles next to those of phase factors would
// spill) only the table position is kept and every use is a fresh scalar load from the constant cache.
template <int N, typename T>
struct Twiddles {
    // not held either with E >= 20 points per thread (N = 40, 80, 96: 15 - 18 complex factors next to the transform's
    // own registers): re-read at every use (scalar loads where a row of threads covers whole waves, see row_pad)
    static constexpr bool HELD = !split_tile<N, T>() && Wg<N>::E < 20;
    static constexpr int TT = Wg<N>::T, Q = Wg<N>::Q;
    cx<T> w[HELD ? (TT - 1) * Q : 1];
    const cx<T>* row;   // tw + 0
    int u;
    template <class Ctx>
    BFSM_HD void load(const cx<T>* tw, int u_, Ctx& ctx) {
        row = tw;
        u = u_;
        if constexpr (HELD) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int uu = 1; uu < TT; ++uu) w[q * (TT - 1) + uu - 1] = ctx.ldc(tw + ((uu * (u_ + TT * q)) % N));
        }
    }
    // twiddle of the value that source thread uu contributes to this thread's q-th sub-transform (uu >= 1)
    template <class Ctx>
    BFSM_HD cx<T> get(int q, int uu, Ctx& ctx) const {
        if constexpr (HELD) return w[q * (TT - 1) + uu - 1];
        else {                                    // opaque: keeps the load where it is used (no hoisting out of the loops)
            const int uo = (Wg<N>::ROW % 64 == 0) ? ctx.opaque(u) : ctx.opaque_v(u);
            return ctx.ldc(row + ((uu * (uo + TT * q)) % N));
        }
    }
};

// LDS exchange discipline.  Every exchange is  [barrier] -> write -> barrier -> read -> [barrier].  Between the reads
// of one exchange and the writes of the next there must be exactly one barrier; where it sits is a scheduling choice:
//   SYNC_POST (behind the reads): a wave that finishes the butterflies between two exchanges stores its results at once,
//             while the slower waves of its SIMD still compute, so the LDS stores (the slow side of the LDS, ~80 B/clk
//             against 256 B/clk for reads) overlap the other waves' arithmetic instead of starting together;
//   SYNC_PRE  (in front of the writes): nothing waits behind the reads, so global loads / stores that follow the last
//             exchange of a tile are issued without a rendezvous.
// fft_tile uses POST inside the tile and leaves the last exchange open (the next tile's first exchange has PRE).
// BFSM_SYNC_BEFORE_WRITE restores PRE everywhere (the round-1 placement).
enum : int { SYNC_PRE = 1, SYNC_POST = 2 };
#ifdef BFSM_SYNC_BEFORE_WRITE
#define BFSM_SYNC_FIX(x) (SYNC_PRE)
#else
#define BFSM_SYNC_FIX(x) (x)
#endif

// step 2 of a line transform on the exchanged values w2[q*T + uu]; result back in v (distribution u + T*m)
template <int N, int SGN, typename T, class Ctx>
BFSM_HD void fft_line_step2(cx<T>* v, cx<T>* w2, const Twiddles<N, T>& twr, Ctx& ctx) {
    constexpr int TT = Wg<N>::T, Q = Wg<N>::Q;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        cx<T> w[TT];
        w[0] = {(T)1, (T)0};
#pragma unroll
        for (int uu = 1; uu < TT; ++uu) w[uu] = twr.get(q, uu, ctx);
        SmallDftTw<TT, SGN, T, true>::run(w2 + q * TT, w);
    }
    // output index k1 + E*k2 with k1 = u + T*q  ==  u + T*(q + Q*k2)
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int k2 = 0; k2 < TT; ++k2) v[q + Q * k2] = w2[q * TT + k2];
}

template <int N, int NP, int SGN, typename T, bool SPLIT = false, int SYNC_ = SYNC_PRE, class Ctx>
BFSM_HD void fft_line_np(cx<T>* v, cx<T>* lds, int p, int u, const Twiddles<N, T>& twr, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T, Q = Wg<N>::Q, LS = NP + 1;   // p in [0, NP): LDS column
    constexpr int SYNC = BFSM_SYNC_FIX(SYNC_);
    SmallDft<E, SGN, T>::run(v);
    cx<T> w2[E];
    if constexpr (!SPLIT) {
        if constexpr ((SYNC & SYNC_PRE) != 0) ctx.sync();
#pragma unroll
        for (int k1 = 0; k1 < E; ++k1) ctx.lds_st(lds + (k1 * TT + u) * LS + p, v[k1]);
        ctx.sync();
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int uu = 0; uu < TT; ++uu) w2[q * TT + uu] = ctx.lds_ld(lds + ((u + TT * q) * TT + uu) * LS + p);
        if constexpr ((SYNC & SYNC_POST) != 0) ctx.sync();
    } else {   // real parts, then imaginary parts, through a scalar buffer (see split_tile)
        T* ls = reinterpret_cast<T*>(lds);
        if constexpr ((SYNC & SYNC_PRE) != 0) ctx.sync();
#pragma unroll
        for (int k1 = 0; k1 < E; ++k1) ls[(k1 * TT + u) * LS + p] = v[k1].x;
        ctx.sync();
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int uu = 0; uu < TT; ++uu) w2[q * TT + uu].x = ctx.lds_ld_s(ls + ((u + TT * q) * TT + uu) * LS + p);
        ctx.sync();
#pragma unroll
        for (int k1 = 0; k1 < E; ++k1) ls[(k1 * TT + u) * LS + p] = v[k1].y;
        ctx.sync();
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int uu = 0; uu < TT; ++uu) w2[q * TT + uu].y = ctx.lds_ld_s(ls + ((u + TT * q) * TT + uu) * LS + p);
        if constexpr ((SYNC & SYNC_POST) != 0) ctx.sync();
    }
    fft_line_step2<N, SGN, T>(v, w2, twr, ctx);
}

template <int N, int SGN, typename T, int SYNC_ = SYNC_PRE, bool SPLIT = split_tile<N, T>(), class Ctx>
BFSM_HD void fft_line(cx<T>* v, cx<T>* lds, int p, int u, const Twiddles<N, T>& twr, Ctx& ctx) {
    fft_line_np<N, N, SGN, T, SPLIT, SYNC_>(v, lds, p, u, twr, ctx);
}

// ---- 2-D transform of an N x N tile with transposition -------------------------------------------------------
// entry: v[m] = tile[a = u + T*m][c = p]      (c is the contiguous axis of the source)
// exit : v[m] = TILE[a' = p][c' = u + T*m]    (to be stored as out[c'][a'], a' contiguous)
// LAST_POST: close the last exchange with a barrier behind its reads (pays when arithmetic or stores follow: KA) or leave
// it open for the next tile's first exchange (pays when the next tile's global loads follow: KC at two workgroups per CU).
// SPLIT: exchange the real and the imaginary parts one after the other through a buffer of scalars (where the complex
// tile does not fit the LDS).
template <int N, int SGN, typename T, bool LAST_POST = false, bool SPLIT = split_tile<N, T>(), class Ctx>
BFSM_HD void fft_tile(cx<T>* v, cx<T>* lds, int p, int u, const Twiddles<N, T>& twr, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T, LS = Wg<N>::LS;
    // the split-exchange geometry keeps the round-1 placement (measured: the new one is slower there)
    constexpr bool OLD = SPLIT;
    constexpr int SYNC = OLD ? SYNC_PRE : BFSM_SYNC_FIX(SYNC_POST);   // the transposing exchange
    fft_line<N, SGN, T, OLD ? SYNC_PRE : (SYNC_PRE | SYNC_POST), SPLIT>(v, lds, p, u, twr, ctx);  // along a
    if constexpr (!SPLIT) {
        if constexpr ((SYNC & SYNC_PRE) != 0) ctx.sync();
#pragma unroll
        for (int m = 0; m < E; ++m) ctx.lds_st(lds + (u + TT * m) * LS + p, v[m]);  // row a', column c
        ctx.sync();
#pragma unroll
        for (int m = 0; m < E; ++m) v[m] = ctx.lds_ld(lds + p * LS + (u + TT * m));  // lane = a', own c = u + T*m
        if constexpr ((SYNC & SYNC_POST) != 0) ctx.sync();
    } else {
        T* ls = reinterpret_cast<T*>(lds);
        if constexpr ((SYNC & SYNC_PRE) != 0) ctx.sync();
#pragma unroll
        for (int m = 0; m < E; ++m) ls[(u + TT * m) * LS + p] = v[m].x;
        ctx.sync();
#pragma unroll
        for (int m = 0; m < E; ++m) v[m].x = ctx.lds_ld_s(ls + p * LS + (u + TT * m));
        ctx.sync();
#pragma unroll
        for (int m = 0; m < E; ++m) ls[(u + TT * m) * LS + p] = v[m].y;
        ctx.sync();
#pragma unroll
        for (int m = 0; m < E; ++m) v[m].y = ctx.lds_ld_s(ls + p * LS + (u + TT * m));
        if constexpr ((SYNC & SYNC_POST) != 0) ctx.sync();
    }
    fft_line<N, SGN, T, OLD ? SYNC_PRE : (LAST_POST ? SYNC_POST : 0), SPLIT>(v, lds, p, u, twr, ctx);  // along c
}

// KC's exchange form.  (Measured at N = 96 in double precision, where the complex tile admits one 6-wave workgroup per
// CU: exchanging the real and imaginary parts separately through 75 KiB so that two workgroups share a CU does not
// speed up its streaming loop -- 1.70 against 1.65 ms for 384 directions -- so KC keeps the geometry's own form.)
template <int N, typename T> constexpr bool kc_split() { return split_tile<N, T>(); }
template <int N, typename T>
constexpr size_t kc_lds_bytes() { return (size_t)Wg<N>::LDS_ELEMS * (kc_split<N, T>() ? sizeof(T) : sizeof(cx<T>)); }

// ------------------------------------------------------------------------------------------------------------
// Kernel parameter blocks (plain pointers and sizes; filled by the host pipeline)
// ------------------------------------------------------------------------------------------------------------
template <typename T>
struct TileFwdRealParams {   // F1a: real f -> 2-D forward transform of every x-plane
    const double* f;         // [x][y][z] real (always double at the API boundary)
    cx<T>* out;              // [x][lz][ly]
    const cx<T>* tw;
};

template <typename T>
struct LineParams {          // generic x-axis pass on an array [x][N*N]
    const cx<T>* in;
    cx<T>* out;
    const cx<T>* tw;
};

template <typename T>
struct GainInvParams {       // KA
    const cx<T>* fhat;       // [lx][lz][ly]
    cx<T>* a1;               // [slot][lx][y][z]; ab_interleaved: [slot][lx][y][z][2] holding both, a2 unused
    cx<T>* a2;
    const cx<T>* phx;        // [shard directions][N] phase tables, exp(i*theta) factors; phx carries the 1/G scale
    const cx<T>* phy;
    const cx<T>* phz;
    const cx<T>* tw;
    long long dir0;          // shard-local index of the chunk's first direction (tables are shard-local)
    int n_dir;               // directions in this chunk
    int per_group;           // directions handled by one workgroup (blockIdx.y)
    size_t a_bstride;        // elements between consecutive batch members (blockIdx.z) in a1 / a2
    int planes;              // lx planes stored per direction: N, or N/2 + 1 (indices 0..N/2) in the Hermitian mode
    int warm_tables;         // != 0: touch the phase-table rows two iterations ahead (tables larger than an XCD's L2)
};

// Hermitian mode where KN's workgroups have KA's shape (nyq_rides_along): the Nyquist-row transforms of the chunk run
// as extra workgroups of KA's launch -- blockIdx.y >= ka_groups -- instead of as a launch of their own
template <typename T>
struct GainInvNyqParams {    // KA + guest KN workgroups
    GainInvParams<T> ka;
    cx<T>* r;                // KN's output (NyqRowsParams::r)
    size_t r_bstride;
    int kn_blocks;           // KN workgroups appended
    int ka_groups;           // blockIdx.y below this: KA
};

template <typename T>
struct GainLineParams {      // KB
    cx<T>* a1;               // in: A1' (ab_interleaved: the pairs {A1', A2'})
    const cx<T>* a2;
    const cx<T>* tw;
    size_t a_bstride;        // batch stride (grid points) of a1 / a2
    cx<T>* pout;             // out: P' [slot][x][y][z] -- a1 itself (in place) unless ab_interleaved
    size_t p_bstride;        // batch stride (elements) of pout
};

// One accumulating workgroup column of KC: a run of directions that all share the radial node r, so that beta1
// (which depends on r and |l|^2 only) can be applied once per slab by body_reduce instead of per direction.
struct Segment {
    int d0;                  // first direction, relative to the chunk
    int n;                   // directions in the segment
    int r;                   // radial node
    int pad;
};

template <typename T>
struct GainFwdParams {       // KC
    const cx<T>* p;          // [slot][x][y][z]
    cx<T>* slab;             // [segment][lx][lz][ly]: sum over the segment of dirw * P_hat (beta1 not yet applied)
    const T* dirw;           // [n_dirs] scalar weight of direction b: (1/G) w_r w_s rho_r^(gamma+2)
    const Segment* segs;     // [n_segments] of the whole plan
    const cx<T>* tw;
    long long dir0;          // shard-local index of the chunk's first direction
    int seg0;                // first segment of this chunk (== its first slab)
    size_t p_bstride;        // batch strides (elements) of p and slab
    size_t slab_bstride;
};

template <typename T>
struct ReduceParams {        // Q_hat[l] = sum_segments beta1[r(seg)][|l|^2] * slab[seg][l]   (fixed order)
    const cx<T>* slab;
    cx<T>* qhat;             // [lx][lz][ly]
    const T* beta1;          // [M_gl][n2stride]: 4 pi b_gamma sincc(pi rho_r sqrt(n2) / (2L))
    const Segment* segs;
    int n_segs;
    int n2stride;            // 3 (N/2)^2 + 1
    size_t slab_bstride;     // batch stride (elements) of slab; qhat is [batch][G]
};

template <typename T>
struct GainLineAccParams {   // KB', exact-reduction mode: sum over the directions of a segment BEFORE the forward FFT
    const cx<T>* a1;         // [slot][lx][y][z]
    const cx<T>* a2;
    cx<T>* pseg;             // [segment][x][y][z]: x-forward transform of sum_d dirw[d] * A1_d * A2_d
    const T* dirw;           // [n_dirs]
    const Segment* segs;
    const cx<T>* tw;
    long long dir0;          // shard-local index of the chunk's first direction
    int seg0;                // first segment of this chunk
    size_t a_bstride;        // batch strides (elements) of a1 / a2 and of pseg
    size_t pseg_bstride;
};

// Hermitian mode (f real): only the planes lx = 0 .. N/2 of A1', A2' are computed and stored.  For idx > N/2
//   A'[idx](y,z) = conj(A'[N - idx](y,z)) + (-1)^y R1[idx](z) + (-1)^z R2[idx](y),
// where R1 / R2 are 1-D inverse transforms of the Nyquist row ly = -N/2 / column lz = -N/2 of
//   phx[idx] * f_hat[idx] / G * (phy (x) phz - phy' (x) phz'),   phy', phz' = the tables with their Nyquist entry
// conjugated (the only modes whose phase is not Hermitian-compatible).  Exact up to rounding.
template <typename T>
struct NyqRowsParams {       // KN: R[slot][sign][kind][j][N], j = idx - (N/2 + 1), kind 0: R1 (over z), 1: R2 (over y)
    const cx<T>* fhat;       // [lx][lz][ly]
    cx<T>* r;
    const cx<T>* phx;
    const cx<T>* phy;
    const cx<T>* phz;
    const cx<T>* tw;
    long long dir0;          // shard-local index of the chunk's first direction
    size_t r_bstride;        // batch stride (elements) of r
};

template <typename T>
struct GainLineAccHParams {  // KB' in the Hermitian mode
    const cx<T>* a1;         // [slot][N/2 + 1 planes][y][z]
    const cx<T>* a2;
    const cx<T>* r;          // Nyquist rows, layout of NyqRowsParams::r
    cx<T>* pseg;             // [segment][x][y][z]
    const T* dirw;
    const Segment* segs;
    const cx<T>* tw;
    long long dir0;
    int seg0;
    size_t a_bstride;
    size_t pseg_bstride;
    size_t r_bstride;
};

template <typename T>
struct TailInvParams {       // tail step 1: plane inverse transforms of Q_hat and beta2 * f_hat
    const cx<T>* qhat;
    const cx<T>* fhat;
    const T* beta2;          // [n2max+1], includes the 1/G scale
    cx<T>* tg;               // [lx][y][z]
    long long tl_off;        // the loss plane's array, as an element offset from tg (tl - tg)
    const cx<T>* tw;
    // n_segs >= 0: the gain plane is formed here from the write-once slabs (the reduce of ReduceParams fused into
    // the load, same summation order, qhat not touched); n_segs < 0: read it from qhat.
    const cx<T>* slab;
    const T* beta1;
    const Segment* segs;
    int n_segs;
    int n2stride;
    size_t slab_bstride;
};

template <typename T>
struct TailLineParams {      // tail step 2: x inverse of both, Q = Re(gain) - Re(loss) * f
    const cx<T>* tg;
    const cx<T>* tl;
    const double* f;
    double* Q;
    const cx<T>* tw;
    int with_loss;           // 0: Q = Re(gain) only (partial result of a rank that does not own the loss term)
};

BFSM_HD int mode_of(int i, int n) { return i < n / 2 ? i : i - n; }

// XCD-aware block index.  Workgroups are handed to the 8 XCDs round-robin by their linear id, and every XCD has its own
// L2.  Where the workgroups of one grid row (same by) re-read the same small table, the (bx, by) pair is re-derived so
// that the workgroups an XCD receives form a contiguous run of the linear order, i.e. whole rows: the table is then
// fetched into one L2 instead of eight.  A bijection of the grid (when its size is a multiple of 8; identity otherwise);
// placement is a speed matter only, any mapping is correct.
template <class Ctx>
BFSM_HD void xcd_rows(Ctx& ctx, int& bx, int& by) {
    const int gx = ctx.gx(), w = gx * ctx.gy();
    bx = ctx.bx();
    by = ctx.by();
#ifndef BFSM_NO_XCD_ROWS
    if (w % 8 == 0) {
        const int id = by * gx + bx, id2 = (id % 8) * (w / 8) + id / 8;
        bx = id2 % gx;
        by = id2 / gx;
    }
#endif
}

template <bool B> struct BoolTag { static constexpr bool value = B; };

// ------------------------------------------------------------------------------------------------------------
// Kernel bodies.  Workgroup = Wg<N>::THREADS threads, tid = u*N + p.
// ------------------------------------------------------------------------------------------------------------

// F1a.  grid.x = N (x planes).  Replaces copy_to_complex (BoltzmannCUDAKernels.cu:4-16) + the (y,z) part of
// cufftExecZ2Z(plan3d, f, f_hat, FORWARD) (CUDABoltzmannOperator.cu:137-140).
template <int N, typename T, class Ctx>
BFSM_HD void body_tile_fwd_real(const TileFwdRealParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    int p, u;
    lane_coords<N, Wg<N>::ROW>(ctx, p, u);
    const int x = ctx.bx();
    const size_t boff = (size_t)ctx.by() * N * N * N;      // batch member
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    cx<T> v[E];
    const double* src = prm.f + boff + (size_t)x * N * N;
#pragma unroll
    for (int m = 0; m < E; ++m) v[m] = {(T)src[(u + TT * m) * N + p], (T)0};
    fft_tile<N, -1, T>(v, lds, p, u, twr, ctx);
    cx<T>* dst = prm.out + boff + (size_t)x * N * N;
#pragma unroll
    for (int m = 0; m < E; ++m) dst[(u + TT * m) * N + p] = v[m];
}

// Generic complex 2-D tile pass (bfsm_fft3d only).  grid = (N planes, batch); in place is allowed because the
// workgroup holds its whole plane in registers before it stores.  in [a][c] -> out [c'][a'].
template <int N, int SGN, typename T, class Ctx>
BFSM_HD void body_tile_c2c(const LineParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    int p, u;
    lane_coords<N, Wg<N>::ROW>(ctx, p, u);
    const size_t base = ((size_t)ctx.by() * N + ctx.bx()) * N * N;
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    cx<T> v[E];
#pragma unroll
    for (int m = 0; m < E; ++m) v[m] = prm.in[base + (u + TT * m) * N + p];
    fft_tile<N, SGN, T>(v, lds, p, u, twr, ctx);
    ctx.sync();  // in place: every thread of the plane has loaded before anyone stores (loads precede the syncs above)
#pragma unroll
    for (int m = 0; m < E; ++m) prm.out[base + (u + TT * m) * N + p] = v[m];
}

// Generic x-axis pass.  grid.x = N*N/NPL (blocks of NPL contiguous columns of the inner N*N space), grid.y = batch.
template <int N, int SGN, typename T, class Ctx>
BFSM_HD void body_line(const LineParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    constexpr int NPL = Wg<N>::NPL;
    int p, u;                                   // p: column inside this block of NPL
    lane_coords<NPL, Wg<N>::LROW>(ctx, p, u);
    const size_t base = (size_t)ctx.by() * N * N * N + (size_t)ctx.bx() * NPL + p;
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    cx<T> v[E];
#pragma unroll
    for (int m = 0; m < E; ++m) v[m] = prm.in[base + (size_t)(u + TT * m) * N * N];
    fft_line_np<N, NPL, SGN, T>(v, lds, p, u, twr, ctx);
#pragma unroll
    for (int m = 0; m < E; ++m) prm.out[base + (size_t)(u + TT * m) * N * N] = v[m];
}

// KA keeps its f_hat plane in registers across the direction loop unless that is more than 64 registers per thread next
// to two tiles' worth of transform data (the split-exchange geometry, and E = 24 points per thread in double precision).
template <int N, typename T> constexpr bool keep_plane() { return !split_tile<N, T>() && Wg<N>::E * sizeof(cx<T>) <= 256; }

// KA processes the two signs of a direction as a software-pipelined pair of tiles at N = 128 in single precision (one
// workgroup per CU: nothing else overlaps its exchanges) and at N = 64 in double precision (two workgroups per CU, and
// still 7 % faster: 12 barriers per direction instead of 14, every burst of LDS stores under the other tile's
// butterflies; 114 VGPRs.  In single precision at N = 64, where four and more workgroups share a CU, it is 7 % slower).  BFSM_NO_PIPELINED_PAIR restores the one-tile-after-the-other form everywhere,
// BFSM_NO_PIPELINED_PAIR_64 at N = 64 only (A/B measurements).  N = 96 (one 384-thread workgroup per CU; the f_hat plane is
// re-read per direction in double precision, 255 VGPRs, no scratch): KA 4.07 -> 3.22 ms in double, 2.18 -> 2.06 ms in single
// precision at 384 directions; at N = 80 it is 4 % slower.  N = 24: KA 0.072 -> 0.048 ms (fp64), 0.039 -> 0.034 (fp32); N = 40:
// -3 %; N = 48: -6 % in single, +1 % in double precision (profiles/r04_other_sizes.txt).
template <int N, typename T> constexpr bool pipelined_pair() {
#ifdef BFSM_NO_PIPELINED_PAIR
    return false;
#elif defined(BFSM_NO_PIPELINED_PAIR_64)
    return N >= 128 && !split_tile<N, T>();
#else
    return (N >= 128 && !split_tile<N, T>()) || (N == 64 && sizeof(T) == 8) || N == 96 || N == 24 || N == 40 || (N == 48 && sizeof(T) == 4);
#endif
}

// Interleaved scratch (round 4).  Where an element is 8 bytes (single precision) A1' and A2' of a grid point are stored
// side by side, [slot][lx][y][z][2], so that KA writes and KB reads BOTH with one 16-byte access per lane: half the
// vector-memory instructions on the CU's one memory pipe, 1-KiB instead of 512-byte runs per wave.  KA finishes tile A's
// last step into registers, runs tile B's last step and stores the pair.  P' cannot overwrite A1' in place then (its
// compact rows belong to other workgroups' pairs): KB writes it to a buffer of its own, [slot][x][y][z].
// Geometry: N = 128 in single precision (config 5), the pipelined-pair form of KA; not in the Hermitian mode, whose line
// kernel rebuilds one array at a time (its launches use the kernel kind GainInvTwo = KA with two arrays; measured with every
// other element of the pairs read instead: KB'H 2.06 -> 2.71 ms on the 768-direction slice).  BFSM_NO_INTERLEAVE restores two arrays everywhere.
template <int N, typename T> constexpr bool ab_interleaved() {
#ifdef BFSM_NO_INTERLEAVE
    return false;
#else
    return sizeof(T) == 4 && N == 128 && pipelined_pair<N, T>();
#endif
}

// A/B build (tools: -DBFSM_KA_XLANE): the LAST line pass of each KA tile at N = 128 fp32 as a wave-private pass.  The
// transposing read hands every wave 8 columns with the 8 threads of a line at lane bits 3..5, the exchange between the two
// register steps is done across lanes (v_permlane32_swap, v_permlane16_swap, DPP row_ror:8: DevCtx::xlane_transpose8) instead
// of through LDS, and the pair loop has 8 barriers instead of 12, none of them next to the global-store burst.  The
// inter-step twiddles are per-lane data then (read from a small LDS table); tile rows get a stride of N + 4 elements so that
// the new transposing read is bank-conflict free.  Measured: DESIGN.md 7.1 / profiles/r04_ka_xlane_lastpass_ab.txt.
template <int N, typename T> constexpr bool ka_xlane() {
#ifdef BFSM_KA_XLANE
    return sizeof(T) == 4 && N == 128 && ab_interleaved<N, T>();
#else
    return false;
#endif
}
template <int N, typename T> constexpr size_t ka_xlane_lds_bytes() { return ((size_t)N * (N + 4) + 8 * 17) * sizeof(cx<T>); }

// (Measured and rejected at N = 128 fp32, profiles/r03_ka_wide32_ab.txt, code in commit ab65758: 32 points per thread, 512 threads per
// tile, the 2-D tile transform in three register passes -- z: 32 x 4, y: 8 x 16 -- with two exchanges and 8 barriers per
// direction instead of three and 12.  Correct, 227 VGPRs, 6.40 against 5.35 ms: half the waves hide less than the exchanges save.)
// (Measured and rejected for the pipelined pair at N = 128 fp32, profiles/r03_ka_tid_exchange_ab.txt, code in commit 1b825c9:
// exchanging through two planes of floats with the M0-relative LDS forms ds_write_addtid_b32 / ds_read_addtid_b32 -- no
// address register, two dwords moved per stored value instead of three.  11 % faster for a line pass in isolation
// (tools/micro/xlane_exchange.hip, modes 5 / 6), no gain in the kernel: 5.40 - 5.49 against 5.35 - 5.36 ms.)
// KA.  grid = (N planes lx, groups).  For each direction of the group and both signs: phase multiply
// (compute_alpha_times_f_hat, BoltzmannCUDAKernels.cu:21-59, with the sincos hoisted into separable tables)
// fused with the (lz,ly) -> (y,z) part of the two batched inverse transforms (CUDABoltzmannOperator.cu:156-164).
// PAIRS: store {A1', A2'} pairs into a1 (ab_interleaved geometries); false there only for the Hermitian mode's launches
// (kernel kind GainInvTwo), which keep two arrays.
template <int N, typename T, bool PAIRS = ab_interleaved<N, T>(), class Ctx>
BFSM_HD void body_gain_inv(const GainInvParams<T>& prm, Ctx& ctx) {
    static_assert(!PAIRS || ab_interleaved<N, T>(), "pair stores need the pipelined-pair form");
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    int p, u;
    lane_coords<N, Wg<N>::ROW>(ctx, p, u);
    const int tid = ctx.tid();
    const int lxi = ctx.bx();
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    // The workgroup's f_hat plane stays in registers across the direction loop, except in the one geometry whose
    // 1024-thread workgroup leaves 128 VGPRs for 2 x 64 of data (N = 128, fp64): there it is re-read every
    // iteration (a 256 KiB plane shared by the workgroups of the plane: L2 / Infinity Cache traffic).
    constexpr bool KEEP = keep_plane<N, T>();
    // re-read geometries: f_hat points in flight per batch (the split-exchange geometry has 4 registers' worth of room;
    // E = 24 has the transform's own registers free at that moment and few waves to hide the L2 latency behind)
    constexpr int REREAD = split_tile<N, T>() ? 4 : 12;
    cx<T> fh[KEEP ? E : 1];
    const size_t bz = (size_t)ctx.bz();
    const cx<T>* src = prm.fhat + bz * N * N * N + (size_t)lxi * N * N;
    if constexpr (KEEP) {
#pragma unroll
        for (int m = 0; m < E; ++m) fh[m] = src[(u + TT * m) * N + p];  // [lz = u + T m][ly = p]
    }
    const int d_begin = ctx.by() * prm.per_group;
    int d_end = d_begin + prm.per_group;
    if (d_end > prm.n_dir) d_end = prm.n_dir;
    if constexpr (pipelined_pair<N, T>()) {
        // N = 128 (one 1024-thread workgroup per CU: no second workgroup to overlap with).  The two signs of a direction
        // are two independent tiles A (alpha) and B (conj alpha); they go through the workgroup's ONE exchange buffer
        // alternately, B one stage behind A, so that every burst of LDS stores drains under the other tile's butterflies
        // instead of in front of a barrier:
        //    dft(A) | wr A, dft(B) | rd A | wr B, step2(A) | rd B | wr^T A, step2(B) | rd^T A | wr^T B, dft(A) | rd^T B |
        //    wr A, dft(B) | rd A | wr B, step2(A), store A | rd B | step2(B), store B
        // ("|" = barrier; 12 per direction instead of 14).  The phase factors are formed once per point and used for both
        // signs (the two tiles are live together here anyway).  Measured against the sequential form: see DESIGN.md 7.1.
        constexpr bool XL = PAIRS && ka_xlane<N, T>();
        constexpr int Q = Wg<N>::Q, LS = XL ? N + 4 : Wg<N>::LS;
        unsigned pl = (unsigned)p * (unsigned)sizeof(cx<T>);   // this lane's byte offset inside a row
        // Exchange addresses.  A tile beyond 64 KiB (N = 128) does not fit the 16-bit immediate offset of the LDS
        // instructions; left to itself the compiler then keeps one address register PER ROW of the upper half (17 VGPRs of
        // addresses in this loop, on a kernel at the 128-VGPR cap).  The rows >= N/2 are therefore addressed from a second
        // base that the optimiser cannot fold back into the first: two registers per access pattern, immediates for the rest.
        constexpr bool BIG = sizeof(cx<T>) * (size_t)Wg<N>::LDS_ELEMS > 65536u;
        constexpr int HALF = N / 2;
        const int hi = BIG ? ctx.opaque_v(HALF * LS) : HALF * LS;
        cx<T>* const wlo = lds + u * LS + p;            // rows k TT + u        (k < E)
        cx<T>* const whi = wlo + hi;
        cx<T>* const rlo = lds + u * TT * LS + p;       // rows (u + TT q) TT + uu
        cx<T>* const rhi = rlo + hi;
        auto xw_line = [&](const cx<T>* v) {
#pragma unroll
            for (int k1 = 0; k1 < E; ++k1) {
                if (k1 * TT < HALF) ctx.lds_st(wlo + k1 * TT * LS, v[k1]);
                else ctx.lds_st(whi + (k1 * TT - HALF) * LS, v[k1]);
            }
        };
        auto xr_line = [&](cx<T>* w2) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int uu = 0; uu < TT; ++uu) {
                    if (q * TT * TT < HALF) w2[q * TT + uu] = ctx.lds_ld(rlo + (q * TT * TT + uu) * LS);
                    else w2[q * TT + uu] = ctx.lds_ld(rhi + (q * TT * TT + uu - HALF) * LS);
                }
        };
        auto xw_tr = [&](const cx<T>* v) { xw_line(v); };      // rows u + T m: the same addresses
        auto xr_tr = [&](cx<T>* v) {
#pragma unroll
            for (int m = 0; m < E; ++m) v[m] = ctx.lds_ld(lds + p * LS + (u + TT * m));
        };
        // last step of a tile: the rows of every radix-T sub-transform are stored as soon as it is done, so the stores of
        // the first sub-transform(s) leave under the arithmetic of the following one instead of in one burst of E
        auto step2_store = [&](cx<T>* base, int d, cx<T>* w2) {
            cx<T>* dst = base + bz * prm.a_bstride + ((size_t)d * prm.planes + lxi) * N * N;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                cx<T> w[TT];
                w[0] = {(T)1, (T)0};
#pragma unroll
                for (int uu = 1; uu < TT; ++uu) w[uu] = twr.get(q, uu, ctx);
                SmallDftTw<TT, +1, T, true>::run(w2 + q * TT, w);
#pragma unroll
                for (int k2 = 0; k2 < TT; ++k2)   // row y = u + T (q + Q k2), z = p
                    ctx.template st_stream_at<Wg<N>::ROW % 64 == 0>(dst + (size_t)(u + TT * (q + Q * k2)) * N, pl, w2[q * TT + k2]);
#ifndef BFSM_KA_STORE_BURST
                ctx.sched_fence();
#endif
            }
        };
        // Interleaved scratch: tile A's last step stays in registers (w2[q T + k2] = row u + T (q + Q k2)) ...
        constexpr bool IL = PAIRS;
        auto step2_keep = [&](cx<T>* w2) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                cx<T> w[TT];
                w[0] = {(T)1, (T)0};
#pragma unroll
                for (int uu = 1; uu < TT; ++uu) w[uu] = twr.get(q, uu, ctx);
                SmallDftTw<TT, +1, T, true>::run(w2 + q * TT, w);
            }
        };
        // ... and leaves with tile B's: one 4 * sizeof(T)-byte store per row and lane, {A1', A2'} of the point
        auto step2_store_pair = [&](int d, const cx<T>* wA, cx<T>* w2, unsigned pl2) {
            cx<T>* dst = prm.a1 + 2 * (bz * prm.a_bstride + ((size_t)d * prm.planes + lxi) * N * N);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                cx<T> w[TT];
                w[0] = {(T)1, (T)0};
#pragma unroll
                for (int uu = 1; uu < TT; ++uu) w[uu] = twr.get(q, uu, ctx);
                SmallDftTw<TT, +1, T, true>::run(w2 + q * TT, w);
#pragma unroll
                for (int k2 = 0; k2 < TT; ++k2)   // row y = u + T (q + Q k2), z = p
                    ctx.template st_stream_pair_at<Wg<N>::ROW % 64 == 0>(dst + 2 * (size_t)(u + TT * (q + Q * k2)) * N, pl2,
                                                                          wA[q * TT + k2], w2[q * TT + k2]);
#ifndef BFSM_KA_STORE_BURST
                ctx.sched_fence();
#endif
            }
        };
#ifdef BFSM_KA_BARRIER_TIMES
        unsigned long long tbar[13] = {}, t_begin = ctx.clk();
#define BFSM_TSYNC(k) { const unsigned long long t0_ = ctx.clk(); ctx.sync(); tbar[k] += ctx.clk() - t0_; }
#else
#define BFSM_TSYNC(k) ctx.sync();
#endif
        // per-lane phy factor and the warm-up touches: wave-uniform row pointer + 32-bit lane offset (no 64-bit address
        // held in VGPRs across the loop: the kernel sits at the 128-VGPR cap)
        constexpr bool ROWU = Wg<N>::ROW % 64 == 0;
        cx<T> py = {(T)0, (T)0};
        if (d_begin < d_end) py = ctx.template ld_at<ROWU>(prm.phy + (size_t)(prm.dir0 + d_begin) * N, (unsigned)p * (unsigned)sizeof(cx<T>));
        if constexpr (XL) {
            // ---- cross-lane form of the last line pass (see ka_xlane) -------------------------------------------------
            // second thread mapping, used behind the transposing exchange: wave w takes the columns z = 8 w + (lane & 7), the
            // 8 threads of a line sit at lane bits 3..5 (u2 = lane >> 3)
            const int lane = tid & 63, u2 = lane >> 3, z2 = (tid >> 6) * 8 + (lane & 7);
            cx<T>* const twl = lds + N * LS;                       // [u2][q][uu] inter-step twiddles, rows of 17 (banks)
            if (tid < 8 * 16) {
                const int tu = tid >> 4, tq = (tid >> 3) & 1, tuu = tid & 7;
                twl[tu * 17 + tq * 8 + tuu] = ctx.ldc(prm.tw + ((tuu * (tu + TT * tq)) % N));
            }
            cx<T>* const trd = lds + z2 * LS + u2;                 // transposing read: row z2, columns u2 + 8 m
            const cx<T>* const tw2 = twl + u2 * 17;
            auto xr_tr2 = [&](cx<T>* v) {
#pragma unroll
                for (int m = 0; m < E; ++m) v[m] = ctx.lds_ld(trd + TT * m);
            };
            // radix-T step behind the cross-lane exchange; results stay in w2[q T + k2] = row y = u2 + T (q + Q k2)
            auto step2_x = [&](cx<T>* w2) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    cx<T> w[TT];
                    w[0] = {(T)1, (T)0};
#pragma unroll
                    for (int uu = 1; uu < TT; ++uu) w[uu] = ctx.lds_ld(tw2 + q * TT + uu);
                    SmallDftTw<TT, +1, T, true>::run(w2 + q * TT, w);
                }
            };
            ctx.drain_loads();
            for (int d = d_begin; d < d_end; ++d) {
                const size_t b = (size_t)(prm.dir0 + d);
                const cx<T> c0 = cmul(ctx.ldc(prm.phx + b * N + lxi), py);
                pl = ctx.lane_off((unsigned)p * (unsigned)sizeof(cx<T>));
                if (d + 1 < d_end) py = ctx.template ld_at<ROWU>(prm.phy + (size_t)(prm.dir0 + d + 1) * N, pl);
                cx<T> va[E], vb[E], wa[E], wb[E];
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const cx<T> ph = cmul(c0, ctx.ldc(prm.phz + b * N + u + TT * m));
                    va[m] = cmul(fh[m], ph);
                    vb[m] = cmulc(fh[m], ph);
                }
                SmallDft<E, +1, T>::run(va);
                BFSM_TSYNC(0);                                        // the previous direction's transposed reads are done
                xw_line(va);
                ctx.sched_fence();
                SmallDft<E, +1, T>::run(vb);
                BFSM_TSYNC(1); xr_line(wa); BFSM_TSYNC(2);
                xw_line(vb);
                ctx.sched_fence();
                fft_line_step2<N, +1, T>(va, wa, twr, ctx);
                BFSM_TSYNC(3); xr_line(wb); BFSM_TSYNC(4);
                xw_tr(va);
                ctx.sched_fence();
                fft_line_step2<N, +1, T>(vb, wb, twr, ctx);
                BFSM_TSYNC(5); xr_tr2(va); BFSM_TSYNC(6);
                xw_tr(vb);
                ctx.sched_fence();
                SmallDft<E, +1, T>::run(va);
                ctx.xlane_transpose8(va);                          // va[q T + uu] = (thread uu's va[u2 + T q])
                step2_x(va);
                BFSM_TSYNC(7); xr_tr2(vb);
                ctx.sched_fence();
                SmallDft<E, +1, T>::run(vb);
                ctx.xlane_transpose8(vb);
                // B's radix-T step and the pair stores: row pointer uniform, lane offset (u2 N + z2) pairs
                cx<T>* dst = prm.a1 + 2 * (bz * prm.a_bstride + ((size_t)d * prm.planes + lxi) * N * N);
                const unsigned pl2 = ctx.lane_off((unsigned)(u2 * N + z2) * (unsigned)(2 * sizeof(cx<T>)));
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    cx<T> w[TT];
                    w[0] = {(T)1, (T)0};
#pragma unroll
                    for (int uu = 1; uu < TT; ++uu) w[uu] = ctx.lds_ld(tw2 + q * TT + uu);
                    SmallDftTw<TT, +1, T, true>::run(vb + q * TT, w);
#pragma unroll
                    for (int k2 = 0; k2 < TT; ++k2)   // row y = u2 + T (q + Q k2), z = z2
                        ctx.template st_stream_pair_at<true>(dst + 2 * (size_t)(TT * (q + Q * k2)) * N, pl2, va[q * TT + k2], vb[q * TT + k2]);
                    ctx.sched_fence();
                }
            }
        } else {
        // no per-iteration vmcnt(0) (= store drain) at the loop header, see DevCtx::drain_loads.  Not at N = 64 in double
        // precision: the write-bound geometry with two workgroups per CU shows no difference either way inside the spread of
        // its two box states (profiles/r04_ka_drain64_ab.txt), so the headline configurations keep their round-3 code
#ifdef BFSM_DRAIN64             // A/B builds (tools only): drain the pre-header loads at N = 64 fp64 as well
        constexpr bool DRAIN = true;
#else
        constexpr bool DRAIN = !(N == 64 && sizeof(T) == 8);
#endif
        if constexpr (DRAIN) ctx.drain_loads();
        for (int d = d_begin; d < d_end; ++d) {
            const size_t b = (size_t)(prm.dir0 + d);
            const cx<T> c0 = cmul(ctx.ldc(prm.phx + b * N + lxi), py);
            pl = ctx.lane_off((unsigned)p * (unsigned)sizeof(cx<T>));     // per-iteration copy, see DevCtx::lane_off
            if (d + 1 < d_end) py = ctx.template ld_at<ROWU>(prm.phy + (size_t)(prm.dir0 + d + 1) * N, pl);
            T warm = (T)0, warm2 = (T)0;     // L2 warm-up of the phase-table rows two directions ahead (see the sequential form below)
            constexpr int EPL = 64 / (int)sizeof(cx<T>), LINES = N / EPL;
            const bool warming = (N % 64 == 0) && 2 * LINES <= 64 && prm.warm_tables != 0 && u == 0 && d + 2 < d_end;
            if (warming) {
                const size_t bw = (size_t)(prm.dir0 + d + 2);
                const unsigned lo = ctx.lane_off((unsigned)(tid % LINES) * 64u);      // one 64-byte line per lane
                warm = ctx.template ld_real_at<ROWU>(prm.phz + bw * N, lo);
                warm2 = ctx.template ld_real_at<ROWU>(prm.phy + bw * N, lo);
            }
            cx<T> va[E], vb[E], wa[E], wb[E];
            // Phase factors formed once for both signs: 4 operations per point fewer, a second tile's worth of registers live
            // across the first transform.  N = 128 fp32 (VALU-bound; 120 VGPRs since the exchange addresses take two registers
            // per pattern): KA 5.12 -> 4.94 ms on the 768-direction slice of config 5.  N = 64 fp64 (124 VGPRs): 0.955 -> 0.967 ms
            // at config 3, so not there (profiles/r04_ka_interleave_ab.txt).  BFSM_KA_SHARE_PHASE=0 / 1 forces it off / on.
#ifdef BFSM_KA_SHARE_PHASE
            constexpr bool SHARE = (BFSM_KA_SHARE_PHASE) != 0;
#else
            constexpr bool SHARE = sizeof(T) == 4 && N == 128;
#endif
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const cx<T> ph = cmul(c0, ctx.ldc(prm.phz + b * N + u + TT * m));
                cx<T> fm;
                if constexpr (KEEP) fm = fh[m]; else fm = src[(u + TT * m) * N + p];   // plane not held: re-read (L2)
                va[m] = cmul(fm, ph);           // alpha1 f_hat / G
                if constexpr (SHARE) vb[m] = cmulc(fm, ph);          // conj(alpha1) f_hat / G
            }
            SmallDft<E, +1, T>::run(va);
            BFSM_TSYNC(0);                         // the previous direction's last exchange has been read
            xw_line(va);
            ctx.sched_fence();
            if constexpr (!SHARE) {
                const cx<T> c0s = ctx.opaque_cx(c0);
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    cx<T> fm;
                    if constexpr (KEEP) fm = fh[m]; else fm = src[(u + TT * m) * N + p];
                    vb[m] = cmulc(fm, cmul(c0s, ctx.ldc(prm.phz + b * N + u + TT * m)));
                }
            }
            SmallDft<E, +1, T>::run(vb);
            BFSM_TSYNC(1); xr_line(wa); BFSM_TSYNC(2);
            xw_line(vb);
            ctx.sched_fence();
            fft_line_step2<N, +1, T>(va, wa, twr, ctx);
            BFSM_TSYNC(3); xr_line(wb); BFSM_TSYNC(4);
            xw_tr(va);
            ctx.sched_fence();
            fft_line_step2<N, +1, T>(vb, wb, twr, ctx);
            BFSM_TSYNC(5); xr_tr(va); BFSM_TSYNC(6);
            xw_tr(vb);
            ctx.sched_fence();
            SmallDft<E, +1, T>::run(va);
            BFSM_TSYNC(7); xr_tr(vb); BFSM_TSYNC(8);
            xw_line(va);
            ctx.sched_fence();
            SmallDft<E, +1, T>::run(vb);
            BFSM_TSYNC(9); xr_line(wa); BFSM_TSYNC(10);
            xw_line(vb);
            ctx.sched_fence();
            if constexpr (IL) step2_keep(wa); else step2_store(prm.a1, d, wa);
            BFSM_TSYNC(11); xr_line(wb);
            ctx.sched_fence();
            if constexpr (IL) step2_store_pair(d, wa, wb, ctx.lane_off((unsigned)p * (unsigned)(2 * sizeof(cx<T>))));
            else step2_store(prm.a2, d, wb);
            if (warming) { ctx.keep_alive(warm); ctx.keep_alive(warm2); }
        }
        }   // !XL
#ifdef BFSM_KA_BARRIER_TIMES
#pragma unroll
        for (int k = 0; k < 13; ++k) ctx.dbg_add(k, tbar[k]);
        ctx.dbg_add(13, ctx.clk() - t_begin);      // the wave's whole direction loop
        ctx.dbg_add(14, 1);                        // waves
#endif
#undef BFSM_TSYNC
    } else if constexpr (N >= 64 && KEEP) {
        // One iteration = one direction; both signs are produced by the same code with the sign a compile-time flag
        // (e^{+-i theta} / G = (phx[lx] * phy[ly = p]) * phz[lz = u + T m], conjugated for sign 1), so no per-point select
        // is executed.  The phz / phx factors are wave-uniform scalar loads; the per-lane phy factor of the NEXT direction
        // is fetched before this direction's transforms so that its latency hides behind the butterflies.
        cx<T> py = {(T)0, (T)0};
        if (d_begin < d_end) py = prm.phy[(size_t)(prm.dir0 + d_begin) * N + p];
        ctx.drain_loads();     // see DevCtx::drain_loads: no store drain at the top of every iteration
        for (int d = d_begin; d < d_end; ++d) {
            const size_t b = (size_t)(prm.dir0 + d);
            const cx<T> c0 = cmul(ctx.ldc(prm.phx + b * N + lxi), py);
            if (d + 1 < d_end) py = prm.phy[(size_t)(prm.dir0 + d + 1) * N + p];
            // L2 warm-up of the phase-table rows of the next direction.  The tables are read with ordinary loads and
            // stay in L2 next to the nontemporal streams as long as they fit (<= ~4 MiB per XCD); for larger direction
            // sets (config 5 on one GPU: 3 x 5.9 MiB) every row would be cold at the top of its iteration and stall all
            // waves of the workgroup (measured: KA 8.7 -> 11 us per direction; with the warm-up 9.0 for any table size).
            // One wave touches each 64-byte line of the phz / phy rows one direction ahead; the values are only kept alive
            // until the first transform is done.  The host enables it where it pays: one workgroup per CU (N = 128) and
            // tables beyond 3 MiB; with two workgroups per CU (N = 64) the other workgroup already covers the miss.
            T warm = (T)0;
            constexpr int EPL = 64 / (int)sizeof(cx<T>), LINES = N / EPL;       // entries per line, lines per row
            const bool warming = (N % 64 == 0) && 2 * LINES <= 64 && prm.warm_tables != 0 && u == 0 && d + 2 < d_end;
            if (warming) {
                const size_t bw = (size_t)(prm.dir0 + d + 2);
                const int l = tid % 64;
                const cx<T>* row = (l < LINES) ? prm.phz : prm.phy;
                warm = row[bw * N + (size_t)(l % LINES) * EPL].x;
            }
            auto one_sign = [&](auto conj_tag) {
                constexpr bool CONJ = decltype(conj_tag)::value;
                // the second sign re-forms its phase factors from an opaque copy of c0: sharing them between the signs
                // would keep a whole extra tile of values alive across the first transform (registers: spills;
                // measured with the factors shared: KA 1.09 -> 2.05 ms at config 3).  Fetching the phz factors one
                // direction ahead was measured too: no difference.
                const cx<T> c0s = CONJ ? ctx.opaque_cx(c0) : c0;
                cx<T> v[E];
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const cx<T> ph = cmul(c0s, ctx.ldc(prm.phz + b * N + u + TT * m));
                    cx<T> fm;
                    if constexpr (KEEP) fm = fh[m]; else fm = src[(u + TT * m) * N + p];
                    v[m] = CONJ ? cmulc(fm, ph) : cmul(fm, ph);   // conj(alpha1) f_hat / G  :  alpha1 f_hat / G
                    if constexpr (!KEEP) { if ((m % REREAD) == REREAD - 1) ctx.sched_fence(); }   // at most REREAD re-read points in flight
                }
                fft_tile<N, +1, T, true>(v, lds, p, u, twr, ctx);
                cx<T>* dst = (CONJ ? prm.a2 : prm.a1) + bz * prm.a_bstride + ((size_t)d * prm.planes + lxi) * N * N;
                const unsigned pl = ctx.lane_off((unsigned)p * (unsigned)sizeof(cx<T>));    // per iteration: see DevCtx::lane_off
#pragma unroll
                for (int m = 0; m < E; ++m)   // [y = u + T m][z = p]
                    ctx.template st_stream_at<Wg<N>::ROW % 64 == 0>(dst + (size_t)(u + TT * m) * N, pl, v[m]);
            };
            one_sign(BoolTag<false>{});
            if (warming) ctx.keep_alive(warm);
            one_sign(BoolTag<true>{});
        }
    } else {
        // small grids (per-lane tables live in VGPRs) and the split-exchange geometry (N = 128, fp64: at the register
        // limit): one loop body with a run-time sign
        // One iteration = one signed direction (j = 2*d + sign).  Phase of this thread's points:
        // e^{+-i theta} / G = (phx[lx] * phy[ly = p]) * phz[lz = u + T m], conjugated for sign 1.  The phz / phx factors
        // are wave-uniform scalar loads; the per-lane phy factor of the NEXT iteration is fetched before this
        // iteration's transform so that its latency hides behind the butterflies.
        const int j_end = 2 * d_end;
        cx<T> py = {(T)0, (T)0};
        if (d_begin < d_end) py = prm.phy[(size_t)(prm.dir0 + d_begin) * N + p];
        ctx.drain_loads();
        for (int j = 2 * d_begin; j < j_end; ++j) {
            const int d = j >> 1;
            const bool conj = (j & 1) != 0;
            const size_t b = (size_t)(prm.dir0 + d);
            const cx<T> c0 = cmul(ctx.ldc(prm.phx + b * N + lxi), py);
            if (j + 1 < j_end) py = prm.phy[(size_t)(prm.dir0 + ((j + 1) >> 1)) * N + p];
            cx<T> v[E];
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const cx<T> ph = cmul(c0, ctx.ldc(prm.phz + b * N + u + TT * m));
                cx<T> fm;
                if constexpr (KEEP) fm = fh[m]; else fm = src[(u + TT * m) * N + p];
                v[m] = conj ? cmulc(fm, ph) : cmul(fm, ph);   // conj(alpha1) f_hat / G  :  alpha1 f_hat / G
                if constexpr (!KEEP) { if ((m % REREAD) == REREAD - 1) ctx.sched_fence(); }   // at most REREAD re-read points in flight
            }
            // L2 warm-up of the phase-table rows of the direction after next.  The tables are read with ordinary loads and
            // stay in L2 next to the nontemporal streams as long as they fit (<= ~4 MiB per XCD); for larger direction
            // sets (config 5 on one GPU: 3 x 5.9 MiB) every row would be cold at the top of its iteration and stall all
            // waves of the workgroup (measured: KA 8.7 -> 11 us per direction; with the warm-up 9.0 for any table size).
            // One wave touches each 64-byte line of the phz / phy rows two iterations ahead; the values are only kept alive
            // until the transform is done.  The host enables it where it pays: one workgroup per CU (N = 128) and tables
            // beyond 3 MiB; with two workgroups per CU (N = 64) the other workgroup already covers the miss.
            T warm = (T)0;
            constexpr int EPL = 64 / (int)sizeof(cx<T>), LINES = N / EPL;       // entries per line, lines per row
            const bool warming = (N % 64 == 0) && 2 * LINES <= 64 && prm.warm_tables != 0 && u == 0 && j + 2 < j_end && (j & 1) == 0;
            if (warming) {
                const size_t bw = (size_t)(prm.dir0 + ((j + 2) >> 1));
                const int l = tid % 64;
                const cx<T>* row = (l < LINES) ? prm.phz : prm.phy;
                warm = row[bw * N + (size_t)(l % LINES) * EPL].x;
            }
            fft_tile<N, +1, T>(v, lds, p, u, twr, ctx);
            if (warming) ctx.keep_alive(warm);
            cx<T>* dst = (conj ? prm.a2 : prm.a1) + bz * prm.a_bstride + ((size_t)d * prm.planes + lxi) * N * N;
            const unsigned pl = ctx.lane_off((unsigned)p * (unsigned)sizeof(cx<T>));
#pragma unroll
            for (int m = 0; m < E; ++m)   // [y = u + T m][z = p]
                ctx.template st_stream_at<Wg<N>::ROW % 64 == 0>(dst + (size_t)(u + TT * m) * N, pl, v[m]);
        }
    }
}

// KA at N = 32 on a PAIR of tiles.  A 32 x 32 tile gives 128 threads = rows of 32 lanes, so a wave straddles two values of
// u and twiddles / phz factors would be per-lane data.  The two signs of a direction are two tiles with the same f_hat
// plane and conjugate phases: they are processed side by side as one 32 x 64 panel -- lanes 0..31 the alpha tile, lanes
// 32..63 the conj(alpha) tile -- by 256 threads, so every wave has ONE u: twiddles and phz factors are scalar loads into
// SGPRs as at N >= 64, and the sign costs one multiply per point instead of a select.  The line transforms treat the 64
// columns independently; the transposing exchange transposes each half on its own.  Same parameters, grid and results.
template <int N> constexpr bool pair_tile() {
#ifdef BFSM_NO_PAIR_TILE
    return false;
#else
    return N == 32;
#endif
}
template <int N> constexpr int pair_threads() { return 2 * N * Wg<N>::T; }
template <int N, typename T> constexpr size_t pair_lds_bytes() { return (size_t)N * (2 * N + 1) * sizeof(cx<T>); }

template <int N, typename T, class Ctx>
BFSM_HD void body_gain_inv_pair(const GainInvParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T, NP = 2 * N, LS = NP + 1;
    const int tid = ctx.tid(), lane = tid % NP, u = ctx.uniform(tid / NP, NP);
    const int h = lane / N, p = lane % N;                  // h: 0 = alpha tile (A1'), 1 = conj(alpha) tile (A2')
    const T sg = h ? (T)-1 : (T)1;
    const int lxi = ctx.bx();
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    const size_t bz = (size_t)ctx.bz();
    const cx<T>* src = prm.fhat + bz * N * N * N + (size_t)lxi * N * N;
    cx<T> fh[E];
#pragma unroll
    for (int m = 0; m < E; ++m) fh[m] = src[(u + TT * m) * N + p];  // [lz = u + T m][ly = p], both halves the same plane
    const int d_begin = ctx.by() * prm.per_group;
    int d_end = d_begin + prm.per_group;
    if (d_end > prm.n_dir) d_end = prm.n_dir;
    cx<T> py = {(T)0, (T)0};
    if (d_begin < d_end) py = prm.phy[(size_t)(prm.dir0 + d_begin) * N + p];
    ctx.drain_loads();
    for (int d = d_begin; d < d_end; ++d) {
        const size_t b = (size_t)(prm.dir0 + d);
        cx<T> c0 = cmul(ctx.ldc(prm.phx + b * N + lxi), py);
        c0.y *= sg;                                        // conj(c0 * phz) = conj(c0) * conj(phz): both factors get the sign
        if (d + 1 < d_end) py = prm.phy[(size_t)(prm.dir0 + d + 1) * N + p];
        cx<T> v[E];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            cx<T> z = ctx.ldc(prm.phz + b * N + u + TT * m);       // wave-uniform: scalar load
            z.y *= sg;
            v[m] = cmul(fh[m], cmul(c0, z));               // alpha f_hat / G  |  conj(alpha) f_hat / G
        }
        fft_line_np<N, NP, +1, T, false, SYNC_PRE | SYNC_POST>(v, lds, lane, u, twr, ctx);   // along lz, 64 columns
#pragma unroll
        for (int m = 0; m < E; ++m) ctx.lds_st(lds + (u + TT * m) * LS + lane, v[m]);          // row a', column (h, c)
        ctx.sync();
#pragma unroll
        for (int m = 0; m < E; ++m) v[m] = ctx.lds_ld(lds + p * LS + h * N + (u + TT * m));    // own half, transposed
        ctx.sync();
        fft_line_np<N, NP, +1, T, false, SYNC_POST>(v, lds, lane, u, twr, ctx);                 // along ly
        cx<T>* dst = (h ? prm.a2 : prm.a1) + bz * prm.a_bstride + ((size_t)d * prm.planes + lxi) * N * N;
#pragma unroll
        for (int m = 0; m < E; ++m)   // [y = u + T m][z = p]
            ctx.template st_stream_at<false>(dst + (size_t)(u + TT * m) * N, (unsigned)p * (unsigned)sizeof(cx<T>), v[m]);
    }
}

// KB.  grid = (N*N/NPL column blocks, directions of the chunk).  x-part of the two inverse transforms, hadamard_product
// (BoltzmannCUDAKernels.cu:62-74) in registers, x-part of the forward transform (CUDABoltzmannOperator.cu:175-178).
template <int N, typename T, class Ctx>
BFSM_HD void body_gain_line(const GainLineParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    constexpr int NPL = Wg<N>::NPL;
    int p, u;                                   // p: column inside this block of NPL
    lane_coords<NPL, Wg<N>::LROW>(ctx, p, u);
    // wave-uniform row pointers + a 32-bit lane offset: the accesses take the scalar-base addressing form, so no
    // per-access 64-bit address lives in VGPRs
    const size_t dcol = (size_t)ctx.by() * N * N * N + (size_t)ctx.bx() * NPL;
    const size_t ubase = (size_t)ctx.bz() * prm.a_bstride + dcol;
    const unsigned pl = (unsigned)p * (unsigned)sizeof(cx<T>);   // lane offset in bytes
    constexpr bool UNI = Wg<N>::LROW % 64 == 0;
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    cx<T> a[E], b[E];
    if constexpr (ab_interleaved<N, T>()) {        // {A1', A2'} of a point in one access
        const cx<T>* AB = prm.a1 + 2 * ubase;
#pragma unroll
        for (int m = 0; m < E; ++m) ctx.template ld_stream_pair_at<UNI>(AB + 2 * (size_t)(u + TT * m) * N * N, 2 * pl, a[m], b[m]);
    } else {
        const cx<T>* A1 = prm.a1 + ubase;
        const cx<T>* A2 = prm.a2 + ubase;
#pragma unroll
        for (int m = 0; m < E; ++m) a[m] = ctx.template ld_stream_at<UNI>(A1 + (size_t)(u + TT * m) * N * N, pl);
#pragma unroll
        for (int m = 0; m < E; ++m) b[m] = ctx.template ld_stream_at<UNI>(A2 + (size_t)(u + TT * m) * N * N, pl);
    }
    fft_line_np<N, NPL, +1, T, false, SYNC_PRE | SYNC_POST>(a, lds, p, u, twr, ctx);
    fft_line_np<N, NPL, +1, T, false, SYNC_POST>(b, lds, p, u, twr, ctx);
#pragma unroll
    for (int m = 0; m < E; ++m) a[m] = cmul(a[m], b[m]);
    fft_line_np<N, NPL, -1, T, false, 0>(a, lds, p, u, twr, ctx);
    // in place unless the scratch is interleaved: the SAME base pointer as the loads then (geometries without the scalar-base
    // form hold per-lane 64-bit addresses -- a second base doubled them: N = 80 KB 1.71 -> 2.15 ms)
    cx<T>* P = ab_interleaved<N, T>() ? prm.pout + (size_t)ctx.bz() * prm.p_bstride + dcol : prm.a1 + ubase;
#pragma unroll
    for (int m = 0; m < E; ++m) ctx.template st_stream_at<UNI>(P + (size_t)(u + TT * m) * N * N, pl, a[m]);
}

// KB' (exact-reduction mode).  grid = (N rows y, segments).  FFT linearity: beta1 depends on r only and the forward
// transform is linear, so sum_s w_s FFT(A1_s * A2_s) = FFT(sum_s w_s A1_s * A2_s): the products of all directions of
// a segment (same radial node) are summed in registers in PHYSICAL space and transformed forward once
// (exact up to rounding order).
template <int N, typename T, class Ctx>
BFSM_HD void body_gain_line_acc(const GainLineAccParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    constexpr int NPL = Wg<N>::NPL;
    int p, u;                                   // p: column inside this block of NPL
    lane_coords<NPL, Wg<N>::LROW>(ctx, p, u);
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    const Segment seg = prm.segs[prm.seg0 + ctx.by()];
    const size_t row = (size_t)ctx.bx() * NPL;                   // uniform; the lane adds pl bytes
    const unsigned pl = (unsigned)p * (unsigned)sizeof(cx<T>);
    constexpr bool UNI = Wg<N>::LROW % 64 == 0;
    cx<T> acc[E];
#pragma unroll
    for (int m = 0; m < E; ++m) acc[m] = {(T)0, (T)0};
    for (int d = seg.d0; d < seg.d0 + seg.n; ++d) {
        const size_t base = (size_t)ctx.bz() * prm.a_bstride + (size_t)d * N * N * N + row;
        cx<T> a[E], b[E];
        if constexpr (ab_interleaved<N, T>()) {
#pragma unroll
            for (int m = 0; m < E; ++m)
                ctx.template ld_stream_pair_at<UNI>(prm.a1 + 2 * (base + (size_t)(u + TT * m) * N * N), 2 * pl, a[m], b[m]);
        } else {
#pragma unroll
            for (int m = 0; m < E; ++m) a[m] = ctx.template ld_stream_at<UNI>(prm.a1 + base + (size_t)(u + TT * m) * N * N, pl);
#pragma unroll
            for (int m = 0; m < E; ++m) b[m] = ctx.template ld_stream_at<UNI>(prm.a2 + base + (size_t)(u + TT * m) * N * N, pl);
        }
        fft_line_np<N, NPL, +1, T>(a, lds, p, u, twr, ctx);
        fft_line_np<N, NPL, +1, T>(b, lds, p, u, twr, ctx);
        const T w = prm.dirw[prm.dir0 + d];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const cx<T> pr = cmul(a[m], b[m]);
            acc[m].x += w * pr.x;
            acc[m].y += w * pr.y;
        }
    }
    fft_line_np<N, NPL, -1, T>(acc, lds, p, u, twr, ctx);
    const size_t obase = (size_t)ctx.bz() * prm.pseg_bstride + (size_t)(prm.seg0 + ctx.by()) * N * N * N + row;
#pragma unroll
    for (int m = 0; m < E; ++m) ctx.template st_at<UNI>(prm.pseg + obase + (size_t)(u + TT * m) * N * N, pl, acc[m]);
}

// KN (Hermitian mode).  grid = (column blocks, 2 * directions of the chunk, batch).  One column = one (kind, j):
// the 1-D inverse transform of the Nyquist row (kind 0: ly = -N/2, running over lz -> z) or column (kind 1:
// lz = -N/2, running over ly -> y, corner excluded) of plane idx = N/2 + 1 + j, for the sign by & 1.
template <int N, typename T, class Ctx>
BFSM_HD void body_nyq_rows(const NyqRowsParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T, NPL = Wg<N>::NPL, NQ = N / 2 - 1, H = N / 2;
    int p, u;
    lane_coords<NPL, Wg<N>::LROW>(ctx, p, u);
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    const int c = ctx.bx() * NPL + p;              // column
    const bool live = c < 2 * NQ;
    const int kind = live ? c / NQ : 0, j = live ? c % NQ : 0;
    const int idx = H + 1 + j;
    const int dslot = ctx.by() >> 1;
    const bool conj = (ctx.by() & 1) != 0;
    const size_t b = (size_t)(prm.dir0 + dslot);
    const cx<T> px = prm.phx[b * N + idx];
    const cx<T> pyN = ctx.ldc(prm.phy + b * N + H), pzN = ctx.ldc(prm.phz + b * N + H);
    const cx<T>* fh = prm.fhat + (size_t)ctx.bz() * N * N * N + (size_t)idx * N * N;
    cx<T> v[E];
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int n = u + TT * m;
        cx<T> f, br;
        if (kind == 0) {          // row ly = -N/2: f_hat[idx][lz = n][ly = H]
            f = fh[(size_t)n * N + H];
            const cx<T> pz = ctx.ldc(prm.phz + b * N + n);
            const cx<T> pzP = (n == H) ? cx<T>{pz.x, -pz.y} : pz;
            br = csub(cmul(pyN, pz), cmulc(pzP, pyN));          // phyN*phz - conj(phyN)*phz'
        } else {                  // column lz = -N/2: f_hat[idx][lz = H][ly = n], corner belongs to kind 0
            f = fh[(size_t)H * N + n];
            const cx<T> py = ctx.ldc(prm.phy + b * N + n);
            br = (n == H) ? cx<T>{(T)0, (T)0} : cmul(py, cx<T>{(T)0, (T)2 * pzN.y});   // phy*(phzN - conj(phzN))
        }
        const cx<T> t = conj ? cmulc(cx<T>{br.x, -br.y}, px) : cmul(px, br);   // conj(px)*conj(br) : px*br
        v[m] = live ? cmul(f, t) : cx<T>{(T)0, (T)0};
    }
    fft_line_np<N, NPL, +1, T>(v, lds, p, u, twr, ctx);
    if (live) {
        cx<T>* dst = prm.r + (size_t)ctx.bz() * prm.r_bstride + ((size_t)ctx.by() * 2 * NQ + (size_t)kind * NQ + j) * N;
#pragma unroll
        for (int m = 0; m < E; ++m) dst[u + TT * m] = v[m];
    }
}

// KN rides along with KA where its workgroups have KA's thread count and one column block covers its 2 (N/2 - 1) columns
// (N = 64: 512 threads both; at N = 128 KA has 1024).  Saves the launch and drain of a 30 us kernel per chunk: KN's
// workgroups take the slots KA's grid leaves free and the slots of KA workgroups that finish early.
template <int N> constexpr bool nyq_rides_along() {
#ifdef BFSM_NO_KN_RIDE
    return false;
#else
    return N == 64 && Wg<N>::THREADS == Wg<N>::LINE_THREADS && 2 * (N / 2 - 1) <= Wg<N>::NPL && !pair_tile<N>();
#endif
}
template <int N, typename T> constexpr size_t gain_inv_lds_bytes() {
    return nyq_rides_along<N>() && line_lds_bytes<N, T>() > tile_lds_bytes<N, T>() ? line_lds_bytes<N, T>() : tile_lds_bytes<N, T>();
}
// block indices seen by a body that runs as a guest of another kernel's grid
template <class Ctx>
struct GuestCtx : Ctx {
    int bx_, by_;
    BFSM_HD GuestCtx(const Ctx& c, int bx, int by) : Ctx(c), bx_(bx), by_(by) {}
    BFSM_HD int bx() const { return bx_; }
    BFSM_HD int by() const { return by_; }
};
template <int N, typename T, class Ctx>
BFSM_HD void body_gain_inv_nyq(const GainInvNyqParams<T>& prm, Ctx& ctx) {
    if (ctx.by() >= prm.ka_groups) {                                     // workgroup-uniform
        const int kn = (ctx.by() - prm.ka_groups) * ctx.gx() + ctx.bx();
        if (kn >= prm.kn_blocks) return;
        const NyqRowsParams<T> pk{prm.ka.fhat, prm.r, prm.ka.phx, prm.ka.phy, prm.ka.phz, prm.ka.tw, prm.ka.dir0, prm.r_bstride};
        GuestCtx<Ctx> g(ctx, 0, kn);
        body_nyq_rows<N, T>(pk, g);
        return;
    }
    body_gain_inv<N, T>(prm.ka, ctx);
}

// x-lines of A1', A2' in the Hermitian mode: rows idx = u + T*m of column col = y*N + z.  Only the planes 0 .. N/2 are
// stored; a row idx > N/2 is the conjugate of the stored row N - idx of the SAME column (plus the exact Nyquist terms
// added by hermitian_line_fix), and that stored row belongs to another thread's (another wave's) share of the line.
// Every stored row is therefore fetched from HBM exactly once, by the thread that owns it (m < E/2, and m == E/2 for
// u == 0: the Nyquist plane), streamed; the threads that need it as a mirror row get it through LDS.  (Round 1 let both
// threads read it from global memory: 1.5x over-fetch measured, the second read missing L1 and often L2.)
// colrow = first column of this workgroup's block (uniform), pl = lane offset in bytes.
template <int N, typename T, class Ctx>
BFSM_HD void hermitian_lines_load(cx<T>* a, cx<T>* b, const cx<T>* A1, const cx<T>* A2, int colrow, unsigned pl, int p,
                                  int u, cx<T>* lds, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T, MS = E / 2, LS = Wg<N>::NPL + 1, H = N / 2;
    constexpr bool UNI = Wg<N>::LROW % 64 == 0;
    static_assert(TT * MS == N / 2, "split row");
    cx<T>* ha = lds;
    cx<T>* hb = lds + (H + 1) * LS;
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        a[m] = ctx.template ld_stream_at<UNI>(A1 + (size_t)(u + TT * m) * N * N + colrow, pl);
        b[m] = ctx.template ld_stream_at<UNI>(A2 + (size_t)(u + TT * m) * N * N + colrow, pl);
    }
    if (u == 0) {                                  // the stored Nyquist plane lx = N/2 is its own mirror
        a[MS] = ctx.template ld_stream_at<UNI>(A1 + (size_t)H * N * N + colrow, pl);
        b[MS] = ctx.template ld_stream_at<UNI>(A2 + (size_t)H * N * N + colrow, pl);
    }
    ctx.sync();                                    // the previous users of the buffer (last exchange) are done
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        ctx.lds_st(ha + (u + TT * m) * LS + p, a[m]);
        ctx.lds_st(hb + (u + TT * m) * LS + p, b[m]);
    }
    ctx.sync();
    // mirror rows: row m is the conjugate of the stored row q = N - (u + T m) = (T - u) + T (E - 1 - m), 1 <= q < N/2.
    // ONE address register (the row of m = E - 1, opaque to the optimiser) + non-negative immediates: left to itself the
    // compiler kept one register per row and array -- 8 of them spilled at the 128-VGPR cap of N = 64 in double precision
    const cx<T>* const mlo = lds + ((N == 32 || N == 64) ? ctx.opaque_v((TT - u) * LS + p) : (TT - u) * LS + p);
#pragma unroll
    for (int m = MS; m < E; ++m) {
        if (m > MS || u != 0) {
            a[m] = ctx.lds_ld(mlo + TT * (E - 1 - m) * LS);
            b[m] = ctx.lds_ld(mlo + ((H + 1) + TT * (E - 1 - m)) * LS);
        }
    }
}

// The same for ONE array (used where the registers do not hold both lines of a direction at once: N = 128).
template <int N, typename T, class Ctx>
BFSM_HD void hermitian_line_load1(cx<T>* a, const cx<T>* A1, int colrow, unsigned pl, int p, int u, cx<T>* lds, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T, MS = E / 2, LS = Wg<N>::NPL + 1, H = N / 2;
    constexpr bool UNI = Wg<N>::LROW % 64 == 0;
#pragma unroll
    for (int m = 0; m < MS; ++m) a[m] = ctx.template ld_stream_at<UNI>(A1 + (size_t)(u + TT * m) * N * N + colrow, pl);
    if (u == 0) a[MS] = ctx.template ld_stream_at<UNI>(A1 + (size_t)H * N * N + colrow, pl);
    ctx.sync();
#pragma unroll
    for (int m = 0; m < MS; ++m) ctx.lds_st(lds + (u + TT * m) * LS + p, a[m]);
    ctx.sync();
#pragma unroll
    for (int m = MS; m < E; ++m)
        if (m > MS || u != 0) a[m] = ctx.lds_ld(lds + (N - (u + TT * m)) * LS + p);
}

template <int N, typename T, class Ctx>
BFSM_HD void hermitian_line_fix(cx<T>* v, const cx<T>* R, int y, int z, int u, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T, NQ = N / 2 - 1, H = N / 2, MS = E / 2;
    const T sy = (y & 1) ? (T)-1 : (T)1, sz = (z & 1) ? (T)-1 : (T)1;
    // Scalar-base form of the Nyquist-row loads (wave-uniform row + one lane offset: no 64-bit address per row in VGPRs)
    // where the kernel sits at the 128-VGPR cap -- N = 32 and 64, where the per-row addresses were what it spilled.  The
    // other geometries keep plain loads, which the compiler schedules early (measured with the scalar-base form everywhere:
    // N = 48 KB'H 0.119 -> 0.150 ms, N = 128 fp32 2.04 -> 2.11 ms; profiles/r04_other_sizes.txt).
    constexpr bool SB = (N == 32 || N == 64) && Wg<N>::LROW % 64 == 0;
    unsigned zoff = 0, yoff = 0;
    if constexpr (SB) zoff = ctx.lane_off((unsigned)z * (unsigned)sizeof(cx<T>));
    if constexpr (SB && Wg<N>::NPL > N) yoff = ctx.lane_off((unsigned)y * (unsigned)sizeof(cx<T>));
#pragma unroll
    for (int m = MS; m < E; ++m) {
        const bool mir = (m > MS) || u != 0;
        const int j = mir ? u + TT * m - (H + 1) : 0;
        // wave-uniform row + one lane offset (scalar-base load: no 64-bit address per row held in VGPRs across the
        // direction loop -- they were what the kernel spilled at the 128-VGPR cap)
        cx<T> r1, r2;
        if constexpr (SB) {
            const int ju = ctx.uniform(j, Wg<N>::LROW);          // (wave-uniform: a row of lanes is whole waves here)
            r1 = ctx.template ld_at<true>(R + (size_t)ju * N, zoff);
            if constexpr (Wg<N>::NPL > N) r2 = ctx.template ld_at<true>(R + (size_t)(NQ + ju) * N, yoff);   // N = 32: y differs inside a wave
            else r2 = ctx.ldc(R + (size_t)(NQ + ju) * N + y);   // (j, y) are wave-uniform
        } else {
            r1 = R[(size_t)j * N + z];
            r2 = ctx.ldc(R + (size_t)(NQ + j) * N + y);          // (j, y) are wave-uniform for N >= 64
        }
        const T k = mir ? (T)1 : (T)0, c = mir ? (T)-1 : (T)1;
        v[m] = {v[m].x + k * (sy * r1.x + sz * r2.x), c * v[m].y + k * (sy * r1.y + sz * r2.y)};
    }
}

// (Measured and rejected, profiles/r03_acch_restructure_ab.txt, code in commit 2cbc66e: fetching the next direction's rows ahead
// at one workgroup per CU -- 0.55 against 0.42 ms at config 3 -- and sending the two lines of a direction through the
// exchange buffer as a pipelined pair at two workgroups per CU -- a tie.  One workgroup needs 2.9 us of LDS stores, butterflies
// and barriers per direction with every global latency hidden; two workgroups overlap that 1.35 x.)
// KB' (Hermitian mode).  Same as body_gain_line_acc, but the x-lines are rebuilt from the stored half.
template <int N, typename T, class Ctx>
BFSM_HD void body_gain_line_acc_h(const GainLineAccHParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T, NPL = Wg<N>::NPL, NQ = N / 2 - 1, NH = N / 2 + 1;
    int p, u;
    lane_coords<NPL, Wg<N>::LROW>(ctx, p, u);
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    // the workgroups of a segment share the Nyquist rows R of its directions (31 KiB per direction and array at N = 64):
    // keep them on one XCD (measured: 1.20 x -> 1.0x of the stored bytes fetched)
    int gbx, gby;
    xcd_rows(ctx, gbx, gby);
    const Segment seg = prm.segs[prm.seg0 + gby];
    // first column of the block in the inner N * N space; (y, z) of this lane's column.  A block is a piece of one row
    // (y uniform over the workgroup) or, at N = 32, two whole rows (y = y0 + p / N)
    const int colrow = gbx * NPL;
    const int y = NPL <= N ? colrow / N : colrow / N + p / N;
    const int z = NPL <= N ? colrow % N + p : p % N;
    const unsigned pl = (unsigned)p * (unsigned)sizeof(cx<T>);
    cx<T> acc[E];
#pragma unroll
    for (int m = 0; m < E; ++m) acc[m] = {(T)0, (T)0};
    for (int d = seg.d0; d < seg.d0 + seg.n; ++d) {
        const size_t abase = (size_t)ctx.bz() * prm.a_bstride + (size_t)d * NH * N * N;
        const cx<T>* R = prm.r + (size_t)ctx.bz() * prm.r_bstride + (size_t)d * 4 * NQ * N;
        cx<T> a[E], b[E];
#ifdef BFSM_ACCH_ONE_LINE          // A/B builds (tools only): the one-line-at-a-time ordering at every size
        constexpr bool ONE_LINE = true;
#else
        constexpr bool ONE_LINE = N >= 128;
#endif
        if constexpr (ONE_LINE) {      // 16 points per thread: one line at a time keeps the kernel inside 128 VGPRs
            hermitian_line_load1<N, T>(a, prm.a1 + abase, colrow, pl, p, u, lds, ctx);      // (two arrays in this mode)
            hermitian_line_fix<N, T>(a, R, y, z, u, ctx);
            fft_line_np<N, NPL, +1, T>(a, lds, p, u, twr, ctx);
            hermitian_line_load1<N, T>(b, prm.a2 + abase, colrow, pl, p, u, lds, ctx);
            hermitian_line_fix<N, T>(b, R + (size_t)2 * NQ * N, y, z, u, ctx);
            fft_line_np<N, NPL, +1, T>(b, lds, p, u, twr, ctx);
        } else {
            hermitian_lines_load<N, T>(a, b, prm.a1 + abase, prm.a2 + abase, colrow, pl, p, u, lds, ctx);
            hermitian_line_fix<N, T>(a, R, y, z, u, ctx);
            fft_line_np<N, NPL, +1, T>(a, lds, p, u, twr, ctx);
            hermitian_line_fix<N, T>(b, R + (size_t)2 * NQ * N, y, z, u, ctx);
            fft_line_np<N, NPL, +1, T>(b, lds, p, u, twr, ctx);
        }
        const T w = prm.dirw[prm.dir0 + d];
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const cx<T> pr = cmul(a[m], b[m]);
            acc[m].x += w * pr.x;
            acc[m].y += w * pr.y;
        }
    }
    fft_line_np<N, NPL, -1, T>(acc, lds, p, u, twr, ctx);
    const size_t obase = (size_t)ctx.bz() * prm.pseg_bstride + (size_t)(prm.seg0 + gby) * N * N * N + colrow;
    // the lane offset of the final stores is re-derived from the thread id (not kept in a register across the direction
    // loop: at the 128-VGPR cap of N = 64 in double precision that one value went to scratch)
    unsigned pl3 = pl;
    if constexpr (N == 32 || N == 64) {
        const int pr3 = ctx.opaque_v(ctx.tid()) % Wg<N>::LROW;
        pl3 = (unsigned)((Wg<N>::LROW == NPL || pr3 < NPL) ? pr3 : pr3 - (Wg<N>::LROW - NPL)) * (unsigned)sizeof(cx<T>);
    }
#pragma unroll
    for (int m = 0; m < E; ++m)
        ctx.template st_at<Wg<N>::LROW % 64 == 0>(prm.pseg + obase + (size_t)(u + TT * m) * N * N, pl3, acc[m]);
}

// KC.  grid = (N planes x, segments of the chunk).  (y,z) part of the forward transform + the direction sum of
// atomic_tensor_contraction (BoltzmannCUDAKernels.cu:79-123) kept in registers: no atomics, one slab store per
// workgroup.  A segment never straddles a radial node, so beta1 is applied later, once per slab.
// The transform is linear and the weights dirw are scalars, so  sum_d dirw_d FFT_yz(P'_d) = FFT_yz(sum_d dirw_d P'_d):
// the workgroup streams every P'_d of its segment exactly once (the same 1*G*c bytes per direction as transforming each
// of them), accumulates the weighted sum in registers and transforms it once.  The streaming loop is then a pure
// HBM read (32 FMAs per tile); rounding order differs from the per-direction form at the 1e-16 (fp64) level.
// BFSM_KC_PER_DIRECTION restores one (y,z) transform per direction (not possible in the split-exchange geometry).
template <int N, typename T, class Ctx>
BFSM_HD void body_gain_fwd(const GainFwdParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    int p, u;
    lane_coords<N, Wg<N>::ROW>(ctx, p, u);
    const int x = ctx.bx();
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    cx<T> acc[E];
#pragma unroll
    for (int m = 0; m < E; ++m) acc[m] = {(T)0, (T)0};
    const Segment seg = prm.segs[prm.seg0 + ctx.by()];
#ifdef BFSM_KC_PER_DIRECTION
    constexpr bool PER_DIRECTION = !kc_split<N, T>();
#else
    constexpr bool PER_DIRECTION = false;
#endif
    if constexpr (PER_DIRECTION) {
        for (int d = seg.d0; d < seg.d0 + seg.n; ++d) {
            const cx<T>* src = prm.p + (size_t)ctx.bz() * prm.p_bstride + ((size_t)d * N + x) * N * N;
            cx<T> v[E];
#pragma unroll
            for (int m = 0; m < E; ++m)   // [y = u + T m][z = p]
                v[m] = ctx.template ld_stream_at<Wg<N>::ROW % 64 == 0>(src + (u + TT * m) * N, (unsigned)p * (unsigned)sizeof(cx<T>));
            fft_tile<N, -1, T, (N >= 128)>(v, lds, p, u, twr, ctx);
            const T w = prm.dirw[prm.dir0 + d];
#pragma unroll
            for (int m = 0; m < E; ++m) {
                acc[m].x += w * v[m].x;
                acc[m].y += w * v[m].y;
            }
        }
    } else if constexpr (E >= 20 || (E >= 12 && sizeof(T) == 8)) {
        // Few waves per CU (N = 48 in fp64, N = 80, 96: 3- to 6-wave workgroups, 12 - 24 points per thread; measured: N = 96
        // fp64 KC 1.65 -> 0.85 ms, N = 48 fp64 0.160 -> 0.130 ms; N = 48 fp32 is faster rolled): the rolled loop would issue
        // a direction's loads, drain them all, accumulate, and only then issue the next direction's.  The points are split
        // into two halves with their own registers, and the loads of the next half are issued before the current one is
        // accumulated, so half a direction's loads are in flight at every moment.
        constexpr int H = E / 2;
        auto row_of = [&](int d) { return prm.p + (size_t)ctx.bz() * prm.p_bstride + ((size_t)d * N + x) * N * N; };
        auto load_half = [&](cx<T>* t, const cx<T>* src, int h) {
#pragma unroll
            for (int m = 0; m < H; ++m)
                t[m] = ctx.template ld_stream_at<Wg<N>::ROW % 64 == 0>(src + (u + TT * (h * H + m)) * N, (unsigned)p * (unsigned)sizeof(cx<T>));
        };
        auto add_half = [&](const cx<T>* t, T w, int h) {
#pragma unroll
            for (int m = 0; m < H; ++m) {
                acc[h * H + m].x += w * t[m].x;
                acc[h * H + m].y += w * t[m].y;
            }
        };
        cx<T> t0[H], t1[H];
        const int d_end = seg.d0 + seg.n;
        if (seg.n > 0) load_half(t0, row_of(seg.d0), 0);
        for (int d = seg.d0; d < d_end; ++d) {
            const cx<T>* src = row_of(d);
            const T w = prm.dirw[prm.dir0 + d];
            load_half(t1, src, 1);
            ctx.sched_fence();
            add_half(t0, w, 0);
            if (d + 1 < d_end) load_half(t0, row_of(d + 1), 0);
            ctx.sched_fence();
            add_half(t1, w, 1);
        }
        fft_tile<N, -1, T, false, kc_split<N, T>()>(acc, lds, p, u, twr, ctx);
    } else {
        for (int d = seg.d0; d < seg.d0 + seg.n; ++d) {
            const cx<T>* src = prm.p + (size_t)ctx.bz() * prm.p_bstride + ((size_t)d * N + x) * N * N;
            const T w = prm.dirw[prm.dir0 + d];
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const cx<T> t = ctx.template ld_stream_at<Wg<N>::ROW % 64 == 0>(src + (u + TT * m) * N, (unsigned)p * (unsigned)sizeof(cx<T>));
                acc[m].x += w * t.x;
                acc[m].y += w * t.y;
            }
        }
        fft_tile<N, -1, T, false, kc_split<N, T>()>(acc, lds, p, u, twr, ctx);
    }
    cx<T>* dst = prm.slab + (size_t)ctx.bz() * prm.slab_bstride + ((size_t)(prm.seg0 + ctx.by()) * N + x) * N * N;
#pragma unroll
    for (int m = 0; m < E; ++m) dst[(u + TT * m) * N + p] = acc[m];   // [lz = u + T m][ly = p]
}

// (Measured and rejected at N = 32, profiles/r04_n32_pairs_ab.txt, code in the commit before this note: KC on a PAIR of
// x-planes -- a 32 x 64 panel by 256 threads, one u per wave, scalar-base streams, the counterpart of body_gain_inv_pair --
// runs config 2's KC in 57 - 59 us against 39 us for the 128-thread tiles: half as many workgroups to stream with.)
// Reduce.  One thread per spectral point.  Applies beta1 (the per-point factor of BoltzmannCUDAKernels.cu:113-114)
// and sums the write-once slabs in a fixed order (deterministic; replaces the atomicAdd pair of cu:120-121).
template <int N, typename T, class Ctx>
BFSM_HD void body_reduce(const ReduceParams<T>& prm, Ctx& ctx) {
    const size_t G = (size_t)N * N * N;
    const size_t idx = (size_t)ctx.bx() * ctx.nthreads() + ctx.tid();
    if (idx >= G) return;
    const int ly = (int)(idx % N), lz = (int)((idx / N) % N), lx = (int)(idx / ((size_t)N * N));
    const int mx = mode_of(lx, N), my = mode_of(ly, N), mz = mode_of(lz, N);
    const int n2 = mx * mx + my * my + mz * mz;
    cx<T> q = {(T)0, (T)0};
    const cx<T>* sl = prm.slab + (size_t)ctx.by() * prm.slab_bstride + idx;
    auto add = [&](int c) {
        const cx<T> t = sl[(size_t)c * G];
        const T b1 = prm.beta1[(size_t)prm.segs[c].r * prm.n2stride + n2];
        q.x += b1 * t.x;
        q.y += b1 * t.y;
    };
    if constexpr (N >= 64) {
        // >= 1024 workgroups: enough threads to cover the latency, the rolled loop streams best (cfg3: 12 vs 16 us)
        for (int c = 0; c < prm.n_segs; ++c) add(c);
    } else {
        // small grids: few workgroups and many slabs -- keep eight slab loads in flight (the sum keeps its order)
#pragma unroll 8
        for (int c = 0; c < prm.n_segs; ++c) add(c);
    }
    prm.qhat[(size_t)ctx.by() * G + idx] = q;
}

// Tail 1.  grid = (N planes lx, 2).  y==0: Q_hat plane; y==1: beta2 * f_hat / G
// (compute_beta2_times_f_hat, BoltzmannCUDAKernels.cu:126-159, with beta2 tabulated by |l|^2), then the
// (lz,ly) -> (y,z) part of the two single inverse transforms (CUDABoltzmannOperator.cu:203-212).
template <int N, typename T, class Ctx>
BFSM_HD void body_tail_inv(const TailInvParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    int p, u;
    lane_coords<N, Wg<N>::ROW>(ctx, p, u);
    const int lxi = ctx.bx();
    const bool loss = ctx.by() != 0;
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    cx<T> v[E];
    const size_t pbase = (size_t)ctx.bz() * N * N * N + (size_t)lxi * N * N;
    if (!loss && prm.n_segs >= 0) {
        // Q_hat plane = sum over slabs of beta1[r(slab)][|l|^2] * slab, in the order of body_reduce (bitwise the same)
        const size_t G = (size_t)N * N * N;
        const int mx = mode_of(lxi, N), my = mode_of(p, N);
        const int n2xy = mx * mx + my * my;        // |l|^2 = n2xy + mz^2 is formed at the point of use (an index array of E
#pragma unroll                                     // entries held across the slab loop went to scratch: 24 B per lane)
        for (int m = 0; m < E; ++m) v[m] = {(T)0, (T)0};
        const cx<T>* sl = prm.slab + (size_t)ctx.bz() * prm.slab_bstride + (size_t)lxi * N * N;
#pragma unroll 2
        for (int c = 0; c < prm.n_segs; ++c) {
            const T* b1row = prm.beta1 + (size_t)prm.segs[c].r * prm.n2stride + n2xy;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const cx<T> t = sl[(size_t)c * G + (u + TT * m) * N + p];
                const int mz = mode_of(u + TT * m, N);
                const T b1 = b1row[mz * mz];
                v[m].x += b1 * t.x;
                v[m].y += b1 * t.y;
            }
        }
    } else if (!loss) {
#pragma unroll
        for (int m = 0; m < E; ++m) v[m] = prm.qhat[pbase + (u + TT * m) * N + p];
    } else {
        const int mx = mode_of(lxi, N), my = mode_of(p, N);
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int mz = mode_of(u + TT * m, N);
            const T b2 = prm.beta2[mx * mx + my * my + mz * mz];
            const cx<T> t = prm.fhat[pbase + (u + TT * m) * N + p];
            v[m] = {b2 * t.x, b2 * t.y};
        }
    }
    fft_tile<N, +1, T>(v, lds, p, u, twr, ctx);
    // (the loss plane is addressed as tg + tl_off: a select between the two pointers was lowered to an indexed load from a
    // private copy of the parameter block -- 24 bytes of scratch per lane in every instantiation)
    cx<T>* dst = prm.tg + (loss ? prm.tl_off : 0) + pbase;
#pragma unroll
    for (int m = 0; m < E; ++m) dst[(u + TT * m) * N + p] = v[m];
}

// Tail 2.  grid.x = N rows y.  x-part of both inverse transforms + compute_Q_total
// (BoltzmannCUDAKernels.cu:162-177): Q = Re(Q_gain) - Re(beta2_times_f * f), f real.
template <int N, typename T, class Ctx>
BFSM_HD void body_tail_line(const TailLineParams<T>& prm, Ctx& ctx) {
    constexpr int E = Wg<N>::E, TT = Wg<N>::T;
    constexpr int NPL = Wg<N>::NPL;
    int p, u;                                   // p: column inside this block of NPL
    lane_coords<NPL, Wg<N>::LROW>(ctx, p, u);
    const size_t base = (size_t)ctx.by() * N * N * N + (size_t)ctx.bx() * NPL + p;
    cx<T>* lds = ctx.template lds<cx<T>>();
    Twiddles<N, T> twr;
    twr.load(prm.tw, u, ctx);
    cx<T> g[E], l[E];
#pragma unroll
    for (int m = 0; m < E; ++m) g[m] = prm.tg[base + (size_t)(u + TT * m) * N * N];
    fft_line_np<N, NPL, +1, T>(g, lds, p, u, twr, ctx);
    if (prm.with_loss) {
#pragma unroll
        for (int m = 0; m < E; ++m) l[m] = prm.tl[base + (size_t)(u + TT * m) * N * N];
        fft_line_np<N, NPL, +1, T>(l, lds, p, u, twr, ctx);
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const size_t i = base + (size_t)(u + TT * m) * N * N;
            prm.Q[i] = (double)g[m].x - (double)l[m].x * prm.f[i];
        }
    } else {
#pragma unroll
        for (int m = 0; m < E; ++m) prm.Q[base + (size_t)(u + TT * m) * N * N] = (double)g[m].x;
    }
}


// ------------------------------------------------------------------------------------------------------------
// Whole-direction kernels for N = 16: the 16^3 grid is 64 KiB of complex doubles, so a workgroup keeps
// a whole direction in registers + LDS and an evaluation is TWO launches with no HBM intermediates besides one real
// partial result per workgroup:
//   small_gain  : every workgroup forms f_hat = FFT(f) itself (32 KiB input, no separate launch to wait for), then for
//                 each of its directions: phase multiply, both inverse 3-D transforms (A1 and A2 exchanged together),
//                 product, forward 3-D transform, weighted accumulate (dirw * beta1) in registers; finally it
//                 inverse-transforms its OWN partial Q_hat (the transform is linear) and stores the real part; one
//                 workgroup (an extra one while a CU is free, else workgroup 0) transforms beta2 f_hat / G alongside
//                 and subtracts the loss term from its share;
//   small_reduce: Q = fixed-order sum of the partial results (replaces the atomics of Kernels.cu:120-121 and the
//                 combine of Kernels.cu:162-177).
// 256 threads; thread (i, j) holds a line of 16 points; three layouts of the cube index (a, b, c) -> (a*16 + b)*17 + c:
//   P: thread (x, y)   holds z = k     S: thread (ly, lz) holds lx = k     M: thread (x | lx, z | lz) holds y | ly = k
// All exchanges are bank-conflict free for 16-byte elements (consecutive lanes hit consecutive or 17-apart slots).
// ------------------------------------------------------------------------------------------------------------
constexpr int SMALL_N = 16;
constexpr int SMALL_THREADS = 256;
constexpr int SMALL_CUBE = SMALL_N * SMALL_N * (SMALL_N + 1);       // padded cube, elements
template <typename T>
constexpr size_t small_lds_bytes() { return (size_t)2 * SMALL_CUBE * sizeof(cx<T>); }

template <typename T>
struct SmallGainParams {
    const double* f;         // [x][y][z]
    T* part;                 // [workgroup][x][y][z]: Re IFFT of this workgroup's share of Q_gain_hat; workgroup loss_wg
                             // also subtracts the loss term from its share
    const T* beta2;          // [n2stride], 1/G folded
    int loss_wg;             // workgroup that owns the loss term (-1: nobody, the caller's rank does not own it): a gain
                             // workgroup (it transforms the loss term alongside its own share) or, when the GPU has a
                             // free CU for it, one extra workgroup after the last gain workgroup (index n_gain_wgs)
    int n_gain_wgs;
    const cx<T>* phx;        // [dirs][16]; phx carries the 1/G scale
    const cx<T>* phy;
    const cx<T>* phz;
    const T* dirw;           // [dirs]
    const T* beta1;          // [n_gl][n2stride]
    int n2stride;
    int n_dirs;              // effective directions of this handle's shard
    int per_wg;              // directions per workgroup (contiguous)
    long long dir_begin;     // global index of the shard's first effective direction (radial node = index / sph_eff)
    int sph_eff;
    int sum_first;           // exact-reduction mode: sum the products of a radial run, transform forward once
};

template <typename T>
struct SmallReduceParams {
    const T* part;           // [n_part][G] real
    double* Q;               // [x][y][z]
    int n_part;
};

enum : int { SMALL_P = 0, SMALL_M = 1, SMALL_S = 2 };
template <int LAYOUT>
BFSM_HD int small_slot(int i, int j, int k) {
    if (LAYOUT == SMALL_P) return (i * 16 + j) * 17 + k;      // (x, y, z = k)
    if (LAYOUT == SMALL_M) return (i * 16 + k) * 17 + j;      // (x, y = k, z)
    return (k * 16 + i) * 17 + j;                             // (x = k, y, z)
}

// re-distribute NA arrays from layout FROM to layout TO through LDS (one buffer of SMALL_CUBE elements per array)
template <int FROM, int TO, int NA, typename T, class Ctx>
BFSM_HD void small_exchange(cx<T> (*v)[16], cx<T>* lds, int i, int j, Ctx& ctx) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int k = 0; k < 16; ++k) ctx.lds_st(lds + a * SMALL_CUBE + small_slot<FROM>(i, j, k), v[a][k]);
    ctx.sync();
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int k = 0; k < 16; ++k) v[a][k] = ctx.lds_ld(lds + a * SMALL_CUBE + small_slot<TO>(i, j, k));
    ctx.sync();
}

// 3-D transforms of NA arrays held by the workgroup.  Forward: layout P in, S out.  Backward: S in, P out.
template <int NA, typename T, class Ctx>
BFSM_HD void small_fft_fwd(cx<T> (*v)[16], cx<T>* lds, int i, int j, Ctx& ctx) {
#pragma unroll
    for (int a = 0; a < NA; ++a) SmallDft<16, -1, T>::run(v[a]);       // along z
    small_exchange<SMALL_P, SMALL_M, NA, T>(v, lds, i, j, ctx);
#pragma unroll
    for (int a = 0; a < NA; ++a) SmallDft<16, -1, T>::run(v[a]);       // along y
    small_exchange<SMALL_M, SMALL_S, NA, T>(v, lds, i, j, ctx);
#pragma unroll
    for (int a = 0; a < NA; ++a) SmallDft<16, -1, T>::run(v[a]);       // along x
}
template <int NA, typename T, class Ctx>
BFSM_HD void small_fft_inv(cx<T> (*v)[16], cx<T>* lds, int i, int j, Ctx& ctx) {
#pragma unroll
    for (int a = 0; a < NA; ++a) SmallDft<16, +1, T>::run(v[a]);       // along lx
    small_exchange<SMALL_S, SMALL_M, NA, T>(v, lds, i, j, ctx);
#pragma unroll
    for (int a = 0; a < NA; ++a) SmallDft<16, +1, T>::run(v[a]);       // along ly
    small_exchange<SMALL_M, SMALL_P, NA, T>(v, lds, i, j, ctx);
#pragma unroll
    for (int a = 0; a < NA; ++a) SmallDft<16, +1, T>::run(v[a]);       // along lz
}

// grid.x = workgroups; each owns per_wg consecutive directions of the shard
template <typename T, class Ctx>
BFSM_HD void body_small_gain(const SmallGainParams<T>& prm, Ctx& ctx) {
    const int tid = ctx.tid(), i = tid >> 4, j = tid & 15;
    cx<T>* lds = ctx.template lds<cx<T>>();
    // f_hat in layout S, kept in registers for every direction of this workgroup
    cx<T> fh[1][16];
#pragma unroll
    for (int k = 0; k < 16; ++k) fh[0][k] = {(T)prm.f[(i * 16 + j) * 16 + k], (T)0};     // layout P: (x = i, y = j, z = k)
    small_fft_fwd<1, T>(fh, lds, i, j, ctx);                                             // layout S: (lx = k, ly = i, lz = j)
    const int my = mode_of(i, 16), mz = mode_of(j, 16);
    T* dst = prm.part + (size_t)ctx.bx() * 4096;
    cx<T> acc[16], psum[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc[k] = {(T)0, (T)0}; psum[k] = {(T)0, (T)0}; }
    const int d0 = ctx.bx() * prm.per_wg;
    int d1 = d0 + prm.per_wg;
    if (d1 > prm.n_dirs) d1 = prm.n_dirs;
    if (ctx.bx() >= prm.n_gain_wgs) d1 = d0;                   // the extra workgroup owns no directions
    for (int d = d0; d < d1; ++d) {
        const int r = (int)((prm.dir_begin + d) / prm.sph_eff);
        // e^{+-i theta} / G = phx[lx] * (phy[ly] * phz[lz])   (compute_alpha_times_f_hat, Kernels.cu:21-59)
        const cx<T> c0 = cmul(prm.phy[(size_t)d * 16 + i], prm.phz[(size_t)d * 16 + j]);
        cx<T> a[2][16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const cx<T> ph = cmul(c0, ctx.ldc(prm.phx + (size_t)d * 16 + k));
            a[0][k] = cmul(fh[0][k], ph);
            a[1][k] = cmulc(fh[0][k], ph);
        }
        small_fft_inv<2, T>(a, lds, i, j, ctx);                 // A1, A2 in layout P
        const T w = prm.dirw[d];
        const bool run_ends = !prm.sum_first || d + 1 == d1 || (int)((prm.dir_begin + d + 1) / prm.sph_eff) != r;
#pragma unroll
        for (int k = 0; k < 16; ++k) {                          // hadamard_product (Kernels.cu:62-74), weighted
            const cx<T> pr = cmul(a[0][k], a[1][k]);
            psum[k].x += w * pr.x;
            psum[k].y += w * pr.y;
        }
        if (run_ends) {                                         // workgroup-uniform
            cx<T> (*ps)[16] = reinterpret_cast<cx<T> (*)[16]>(psum);
            small_fft_fwd<1, T>(ps, lds, i, j, ctx);            // layout S
            const T* b1 = prm.beta1 + (size_t)r * prm.n2stride;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int mx = mode_of(k, 16);
                const T b = b1[mx * mx + my * my + mz * mz];
                acc[k].x += b * psum[k].x;
                acc[k].y += b * psum[k].y;
                psum[k] = {(T)0, (T)0};
            }
        }
    }
    // the inverse transform is linear: this workgroup's share goes back to physical space here, the reduce sums reals
    if (ctx.bx() != prm.loss_wg) {
        cx<T> (*ac)[16] = reinterpret_cast<cx<T> (*)[16]>(acc);
        small_fft_inv<1, T>(ac, lds, i, j, ctx);                                         // layout P: (x = i, y = j, z = k)
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[(i * 16 + j) * 16 + k] = acc[k].x;
    } else {
        // the owner of the loss term transforms beta2 * f_hat / G alongside (compute_beta2_times_f_hat, Kernels.cu:126-159;
        // inverse transforms cu:203-212) and combines: share - Re(loss) * f   (compute_Q_total, Kernels.cu:162-177)
        cx<T> v[2][16];
        // f again, fetched BEFORE the transforms so that its latency hides behind them: this workgroup's tail sets the kernel's
        // duration when every CU is busy (256 directions on 256 CUs: 27.0 -> 22.1 us; prefetching the phase and beta1 factors
        // the same way gained nothing and cost six registers -- profiles/r04_cfg1_prefetch_ab.txt)
        double fre[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) fre[k] = prm.f[(i * 16 + j) * 16 + k];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int mx = mode_of(k, 16);
            const T b2 = prm.beta2[mx * mx + my * my + mz * mz];
            v[0][k] = acc[k];
            v[1][k] = {b2 * fh[0][k].x, b2 * fh[0][k].y};
        }
        small_fft_inv<2, T>(v, lds, i, j, ctx);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            dst[(i * 16 + j) * 16 + k] = (T)((double)v[0][k].x - (double)v[1][k].x * fre[k]);
    }
}

// grid.x = 256 workgroups of 256 threads: workgroup = 16 grid points, 16 groups of partial results; the order of the
// additions is fixed (group-local runs first, then the groups in index order)
template <typename T, class Ctx>
BFSM_HD void body_small_reduce(const SmallReduceParams<T>& prm, Ctx& ctx) {
    const int tid = ctx.tid(), g = tid >> 4, pt = tid & 15;
    const size_t idx = (size_t)ctx.bx() * 16 + pt;
    double* lds = ctx.template lds<double>();
    const int per = (prm.n_part + 15) / 16;
    double q = 0;
    int w = g * per, w1 = (g + 1) * per;
    if (w1 > prm.n_part) w1 = prm.n_part;
    for (; w + 16 <= w1; w += 16) {          // sixteen loads in flight (a group's whole share at config 1), added in the same
        T t[16];                             // fixed order
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = prm.part[(size_t)(w + i) * 4096 + idx];
#pragma unroll
        for (int i = 0; i < 16; ++i) q += (double)t[i];
    }
    for (; w + 4 <= w1; w += 4) {
        T t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = prm.part[(size_t)(w + i) * 4096 + idx];
#pragma unroll
        for (int i = 0; i < 4; ++i) q += (double)t[i];
    }
    for (; w < w1; ++w) q += (double)prm.part[(size_t)w * 4096 + idx];
    lds[g * 16 + pt] = q;
    ctx.sync();
    if (g == 0) {
        double s = lds[pt];
#pragma unroll
        for (int gg = 1; gg < 16; ++gg) s += lds[gg * 16 + pt];
        prm.Q[idx] = s;
    }
}

}  // namespace bfsm

#undef BFSM_SYNC_FIX
