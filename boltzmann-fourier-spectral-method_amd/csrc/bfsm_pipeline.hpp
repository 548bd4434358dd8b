// bfsm_pipeline.hpp -- host-side plan (tables, direction chunks) and the launch sequence of one collision
// evaluation.  Pure C++17, no HIP types: the sequence is a template over a `Backend` that owns memory and
// launches the kernel bodies of bfsm_core.hpp.  bfsm_hip.hip provides the HIP backend (the product);
// tests/emu provides a host lock-step backend so the same plan + sequence is unit-tested without a GPU.
//
// What is computed ("the algorithm in one block"; Collisions/FFTWBoltzmannOperator.cpp:147-334):
//   f_hat = FFT(f)
//   for every direction b=(r,s) of this shard:   A1 = IFFT(e^{+i theta} f_hat / G), A2 = IFFT(e^{-i theta} f_hat / G)
//                                                 P_hat = FFT(A1 * A2)
//                                                 Q_hat += (1/G) w_r w_s rho_r^(gamma+2) * beta1(r,|l|) * P_hat
//   Q = Re IFFT(Q_hat) - Re(IFFT(beta2 f_hat / G)) * f
#pragma once
#include <cmath>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bfsm.h"
#include "bfsm_core.hpp"

namespace bfsm {

struct Chunk {
    long long dir0;   // first direction, LOCAL index inside the shard (tables are shard-local)
    int n;            // directions in the chunk (may span several radial nodes)
    int per_group;    // directions per workgroup column of KA
    int seg0;         // first accumulation segment (== first slab) of the chunk
    int n_seg;        // segments of the chunk (grid.y of KC)
};

struct PlanInfo {
    int N = 0;
    int precision = 64;
    int n_gl = 0, n_sph = 0;
    // Work units are EFFECTIVE directions e = r*sph_eff + s.  Reference-faithful mode: sph_eff = n_sph, weight_mult = 1
    // (one unit per quadrature direction).  Exact-reduction mode on an antipodal design: sph_eff = n_sph/2,
    // weight_mult = 2 (direction s and its antipode s + n_sph/2 give the identical product A1*A2).
    int sph_eff = 0;
    int weight_mult = 1;
    bool exact_reductions = false;         // BFSM_FLAG_EXACT_REDUCTIONS
    bool hermitian = false;                // BFSM_FLAG_HERMITIAN: store only the planes lx = 0..N/2 of A1', A2' 
    bool antipodal = false;                // design satisfies sigma_{s+M/2} == -sigma_s and w equal, bit-exactly
    long long full_begin = 0, full_end = 0;  // the shard as given, in full quadrature directions b = r*n_sph + s
    long long dir_begin = 0, dir_end = 0;  // the shard in effective directions
    int max_chunk = 1024;
    int groups = 8;                        // target workgroup columns per x-plane (>= 2 workgroups per CU)
    std::vector<Segment> segs;             // accumulation segments of all chunks, in slab order
    int n2stride = 0;                      // 3*(N/2)^2 + 1
    std::vector<Chunk> chunks;
    int largest_chunk = 0;
    size_t Gtot = 0;                       // grid points when the plan is not a cube of N (size-generic path, N = 0)
    bool gen_plane = false;                // size-generic path: y and z pass fused through LDS (12 instead of 18 array moves)
    bool gen_fused = false;                // size-generic path: plane / x-line / plane-accumulate kernels (6 array moves)
    int gen_slabs = 0;                     // ... and the partial-sum slabs its chunks write
    int gen_moves = 18;                    // size-generic path: array moves per direction of the sequence chosen (6 ... 18)
    size_t G() const { return Gtot ? Gtot : (size_t)N * N * N; }
    long long n_dirs() const { return dir_end - dir_begin; }
};

// sincc(x) = sin(x+eps)/(x+eps)  (Collisions/FFTWBoltzmannOperator.hpp:17-21, BoltzmannCUDAKernels.hpp:17-29)
inline double sincc_ref(double x) {
    const double eps = DBL_EPSILON;
    return std::sin(x + eps) / (x + eps);
}

// Cubic grids of these sizes (N = Q T^2 with T = 4 or 8 threads per line) run on the fused three-kernel pipeline; every
// other supported grid on the size-generic path of bfsm_generic.hpp.
inline bool fused_grid(const bfsm_desc& d) {
    const int N = d.nvx;
    return d.nvx == d.nvy && d.nvx == d.nvz && (N == 16 || N == 24 || N == 32 || N == 40 || N == 48 || N == 64 || N == 80 || N == 96 || N == 128);
}

// An axis length the library has a transform for: even (the reference's mode tables need it,
// FFTWBoltzmannOperator.cpp:50-57), 4 <= n <= 256, prime factors 2, 3, 5, 7, 11, 13.
inline bool axis_supported(int n) {
    if (n < 4 || n > 256 || n % 2 != 0) return false;
    for (int p : {2, 3, 5, 7, 11, 13}) while (n % p == 0) n /= p;
    return n == 1;
}

inline int validate_desc(const bfsm_desc& d, std::string& err) {
    if (!axis_supported(d.nvx) || !axis_supported(d.nvy) || !axis_supported(d.nvz)) {
        err = "every grid extent must be even, in [4, 256], with prime factors 2, 3, 5, 7, 11, 13 only";
        return BFSM_ERR_UNSUPPORTED;
    }
    if (d.precision != BFSM_F64 && d.precision != BFSM_F32) { err = "precision must be BFSM_F64 or BFSM_F32"; return BFSM_ERR_INVALID; }
    if (d.n_gl < 1 || d.n_sph < 1) { err = "n_gl and n_sph must be positive"; return BFSM_ERR_INVALID; }
    if (!d.gl_nodes || !d.gl_wts || !d.sph_wts || !d.sx || !d.sy || !d.sz) { err = "null quadrature array"; return BFSM_ERR_INVALID; }
    if (!(d.L > 0)) { err = "L must be positive"; return BFSM_ERR_INVALID; }
    const long long B = (long long)d.n_gl * d.n_sph;
    if (!(d.dir_begin == 0 && d.dir_end == 0) && (d.dir_begin < 0 || d.dir_end > B || d.dir_begin > d.dir_end)) {
        err = "direction shard out of range"; return BFSM_ERR_INVALID;
    }
    if (d.max_chunk < 0 || d.max_chunk > 16384) { err = "max_chunk must be in [0, 16384] (it is a grid dimension)"; return BFSM_ERR_INVALID; }
    if (d.max_batch < 0 || d.max_batch > 65535) { err = "max_batch must be in [0, 65535]"; return BFSM_ERR_INVALID; }
    if (!fused_grid(d) && d.max_chunk > 16383) { err = "max_chunk too large for this grid"; return BFSM_ERR_INVALID; }
    if ((d.flags & BFSM_FLAG_HERMITIAN) && !(d.flags & BFSM_FLAG_EXACT_REDUCTIONS)) {
        err = "BFSM_FLAG_HERMITIAN is an additional exact reduction: set BFSM_FLAG_EXACT_REDUCTIONS as well";
        return BFSM_ERR_INVALID;
    }
    return BFSM_OK;
}

// Workgroups a plane-parallel kernel (KA, KC) should offer: two 512-thread workgroups per CU at N >= 64 (what the
// 65 KiB tiles allow); the small grids have 128- and 64-thread workgroups, so the same 16 waves per CU take 8 / 16
// workgroups per CU.
// A handle created for batches of nb distributions multiplies every grid by nb, so the per-distribution target shrinks
// accordingly (never below 512: fewer, longer segments mean fewer slabs to write and reduce).
inline int target_workgroups(int N, int max_batch = 1) {
    const int single = N >= 64 ? 512 : (N == 32 ? 2048 : 1024);
    const int nb = max_batch > 1 ? max_batch : 1;
    return single / nb > 512 ? single / nb : 512;
}

inline PlanInfo make_plan(const bfsm_desc& d) {
    PlanInfo p;
    p.N = d.nvx;
    p.precision = d.precision;
    p.n_gl = d.n_gl;
    p.n_sph = d.n_sph;
    const long long B = (long long)d.n_gl * d.n_sph;
    if (d.dir_begin == 0 && d.dir_end == 0) { p.full_begin = 0; p.full_end = B; }
    else { p.full_begin = d.dir_begin; p.full_end = d.dir_end; }
    p.exact_reductions = (d.flags & BFSM_FLAG_EXACT_REDUCTIONS) != 0;
    p.hermitian = p.exact_reductions && (d.flags & BFSM_FLAG_HERMITIAN) != 0;
    p.sph_eff = d.n_sph;
    if (p.exact_reductions && d.n_sph % 2 == 0) {
        const int h = d.n_sph / 2;
        bool anti = true;
        for (int s = 0; s < h && anti; ++s)
            anti = d.sx[s + h] == -d.sx[s] && d.sy[s + h] == -d.sy[s] && d.sz[s + h] == -d.sz[s] &&
                   d.sph_wts[s + h] == d.sph_wts[s];
        if (anti) { p.antipodal = true; p.sph_eff = h; p.weight_mult = 2; }
    }
    // a shard of full directions maps to the proportional range of effective directions (monotone, so the ranks'
    // ranges still tile the whole set exactly once)
    p.dir_begin = p.full_begin * p.sph_eff / d.n_sph;
    p.dir_end = p.full_end * p.sph_eff / d.n_sph;
    // Directions resident at once.  Sized for 288 GB of HBM: by default the whole shard (up to 1024 directions,
    // i.e. 8 GiB of A1'/A2' scratch at N=64 fp64) is one chunk, so an evaluation is ~8 launches.
    p.max_chunk = d.max_chunk > 0 ? d.max_chunk : 1024;
    p.groups = (target_workgroups(p.N, d.max_batch) + p.N - 1) / p.N;
    // KC only streams and sums until its one transform per segment, so at N = 32 half as many, twice as long segments
    // win: half the slabs to write and reduce (config 2: KC 42 -> 40 us, reduce 12 -> 8 us)
    if (p.N == 32) p.groups /= 2;
    if (p.groups < 1) p.groups = 1;
    p.n2stride = 3 * (p.N / 2) * (p.N / 2) + 1;
    const long long len = p.dir_end - p.dir_begin;
    if (len > 0) {
        const long long pieces = (len + p.max_chunk - 1) / p.max_chunk;
        const long long piece = (len + pieces - 1) / pieces;
        for (long long o = 0; o < len; o += piece) {
            Chunk c;
            c.dir0 = o;
            c.n = (int)((o + piece <= len) ? piece : (len - o));
            c.per_group = (c.n + p.groups - 1) / p.groups;
            c.seg0 = (int)p.segs.size();
            // radial runs inside the chunk; every run is cut into `cuts` near-equal segments so that the chunk
            // offers at least `groups` accumulating workgroups per x-plane
            const long long g0 = p.dir_begin + o, g1 = g0 + c.n;
            const int r_first = (int)(g0 / p.sph_eff), r_last = (int)((g1 - 1) / p.sph_eff);
            const int runs = r_last - r_first + 1;
            int cuts = (p.groups + runs - 1) / runs;
            // KC runs N x segments workgroups; with the large tiles only `resident` of them fit the GPU at once
            // (one per CU at N = 128, two at N = 64), so a segment count that is not a multiple of resident / N leaves
            // a partly empty last round (config 5: 5 runs -> 640 workgroups = 2.5 rounds; 10 segments = 5 full rounds)
            const int resident = p.N >= 128 ? 256 : (p.N == 64 ? 512 : 0);
            if (resident > 0 && resident % p.N == 0) {
                const int unit = resident / p.N;
                for (int extra = 0; extra < unit && (runs * cuts) % unit != 0; ++extra) ++cuts;
            } else if (p.N == 48 || p.N == 96 || p.N == 80 || p.N == 24 || p.N == 40) {
                // sizes whose row count does not divide the resident set: among cuts .. cuts + 3 take the count that
                // fills its rounds of resident workgroups best (N = 96 / 80: one / two or three workgroups per CU by LDS;
                // N = 48: four / eight; N = 24: one-wave workgroups, 16 per CU)
                const bool f64 = p.precision == BFSM_F64;
                const int res = p.N == 96 ? (f64 ? 256 : 512) : p.N == 80 ? (f64 ? 256 : 768) : p.N == 48 ? (f64 ? 1024 : 2048) : (p.N == 40 ? 2048 : 4096);
                int best = cuts;
                double best_u = 0;
                for (int c2 = cuts; c2 <= cuts + 3; ++c2) {
                    const long long w = (long long)p.N * runs * c2, rounds = (w + res - 1) / res;
                    const double u = (double)w / (double)(rounds * res);
                    if (u > best_u + 0.02) { best_u = u; best = c2; }
                }
                cuts = best;
            }
            for (int r = r_first; r <= r_last; ++r) {
                long long a0 = (long long)r * p.sph_eff, a1 = a0 + p.sph_eff;
                if (a0 < g0) a0 = g0;
                if (a1 > g1) a1 = g1;
                const long long rl = a1 - a0;
                const long long k = rl < cuts ? rl : cuts;
                for (long long j = 0; j < k; ++j) {
                    const long long s0 = a0 + rl * j / k, s1 = a0 + rl * (j + 1) / k;
                    Segment sg;
                    sg.d0 = (int)(s0 - g0);
                    sg.n = (int)(s1 - s0);
                    sg.r = r;
                    sg.pad = 0;
                    p.segs.push_back(sg);
                }
            }
            c.n_seg = (int)p.segs.size() - c.seg0;
            if (c.n > p.largest_chunk) p.largest_chunk = c.n;
            p.chunks.push_back(c);
        }
    }
    return p;
}

// Host tables, built in double / long double, then narrowed to T.
template <typename T>
struct HostTables {
    std::vector<cx<T>> tw;                 // [N]
    std::vector<cx<T>> phx, phy, phz;      // [n_dirs][N]
    std::vector<T> dirw;                   // [n_dirs]
    std::vector<T> beta1;                  // [n_gl][n2stride]
    std::vector<T> beta2;                  // [n2stride]
};

template <typename T>
HostTables<T> build_tables(const bfsm_desc& d, const PlanInfo& p) {
    HostTables<T> t;
    const int N = p.N;
    const double pi = 3.14159265358979323846;  // Utilities/constants.hpp:7
    const long double PI_L = 3.141592653589793238462643383279502884L;
    const double G = (double)p.G();
    const double fft_scale = 1.0 / G;          // FFTWBoltzmannOperator.cpp:162
    t.tw.resize(N);
    for (int n = 0; n < N; ++n) {
        const long double a = -2.0L * PI_L * (long double)n / (long double)N;
        t.tw[n] = {(T)cosl(a), (T)sinl(a)};
    }
    const long long nd = p.n_dirs();
    t.phx.resize((size_t)nd * N);
    t.phy.resize((size_t)nd * N);
    t.phz.resize((size_t)nd * N);
    t.dirw.resize((size_t)nd);
    for (long long i = 0; i < nd; ++i) {
        const long long b = p.dir_begin + i;
        const int r = (int)(b / p.sph_eff), s = (int)(b % p.sph_eff);   // s < n_sph/2 when antipodal pairs are merged
        // theta(l) = -(pi/(2L)) rho_r (l . sigma_s)   (FFTWBoltzmannOperator.cpp:205-209), separable in lx, ly, lz
        const long double k = -((long double)pi / (2.0L * (long double)d.L)) * (long double)d.gl_nodes[r];
        for (int n = 0; n < N; ++n) {
            const int l = n < N / 2 ? n : n - N;   // FFT-order mode (FFTWBoltzmannOperator.cpp:50-57)
            const long double ax = k * (long double)l * (long double)d.sx[s];
            const long double ay = k * (long double)l * (long double)d.sy[s];
            const long double az = k * (long double)l * (long double)d.sz[s];
            t.phx[(size_t)i * N + n] = {(T)(fft_scale * (double)cosl(ax)), (T)(fft_scale * (double)sinl(ax))};
            t.phy[(size_t)i * N + n] = {(T)cosl(ay), (T)sinl(ay)};
            t.phz[(size_t)i * N + n] = {(T)cosl(az), (T)sinl(az)};
        }
        // weight = fft_scale * gl_wts[r] * spherical_wts[s] * pow(gl_nodes[r], gamma + 2)   (cpp:252)
        t.dirw[(size_t)i] = (T)(p.weight_mult * (fft_scale * d.gl_wts[r] * d.sph_wts[s] * std::pow(d.gl_nodes[r], d.gamma + 2)));
    }
    t.beta1.resize((size_t)d.n_gl * p.n2stride);
    t.beta2.assign((size_t)p.n2stride, (T)0);
    std::vector<double> b2((size_t)p.n2stride, 0.0);
    for (int n2 = 0; n2 < p.n2stride; ++n2) {
        const double norm_l = std::sqrt((double)n2);
        for (int r = 0; r < d.n_gl; ++r) {
            // beta1 = 4 pi b_gamma sincc(pi rho_r |l| / (2L))   (cpp:261-262)
            t.beta1[(size_t)r * p.n2stride + n2] = (T)(4 * pi * d.b_gamma * sincc_ref(pi * d.gl_nodes[r] * norm_l / (2 * d.L)));
            // beta2 += 16 pi^2 b_gamma w_r rho_r^(gamma+2) sincc(pi rho_r |l| / L)   (cpp:290-293)
            b2[n2] += 16 * pi * pi * d.b_gamma * d.gl_wts[r] * std::pow(d.gl_nodes[r], d.gamma + 2) *
                      sincc_ref(pi * d.gl_nodes[r] * norm_l / d.L);
        }
        t.beta2[n2] = (T)(fft_scale * b2[n2]);   // cpp:295-296 folds fft_scale
    }
    return t;
}

// Kernels of the N = 16 whole-direction path (bfsm_core.hpp, "small_*").
enum class SK { Gain, Reduce };

// Kernel identifiers the backend dispatches on.
enum class K { TileFwdReal, LineFwd, LineInv, TileFwd, TileInv, GainInv, GainLine, GainFwd, Reduce, TailInv, TailLine, GainLineAcc,
               NyqRows, GainLineAccH, GainInvNyq,     // appended: the numeric values of the others appear in profiles
               GainInvTwo };                          // KA storing two arrays on a geometry whose default is interleaved pairs

// 1-D (x-axis) kernels take Wg<N>::NPL columns per workgroup, 2-D tile kernels a whole N x N tile.
constexpr bool is_line_kind(K k) {
    return k == K::LineFwd || k == K::LineInv || k == K::GainLine || k == K::GainLineAcc || k == K::TailLine ||
           k == K::NyqRows || k == K::GainLineAccH;
}

// Device-resident state of one handle.  `Backend` supplies:
//   void* alloc(size_t), void release(void*), void upload(void* dst, const void* src, size_t), void zero(void*, size_t)
//   template <K kind, typename T, class P> void launch(int grid_x, int grid_y, int grid_z, const P& params, int N)
//   void mark(int kind, double alg_bytes)              -- accounting tag of the next launch (kind < 0: untracked)
template <typename T, class Backend>
struct Pipeline {
    PlanInfo plan;
    Backend* be = nullptr;
    // device buffers
    cx<T>* fhat = nullptr;
    cx<T>* tg = nullptr;
    cx<T>* tl = nullptr;
    cx<T>* qhat = nullptr;
    cx<T>* a1 = nullptr;
    cx<T>* a2 = nullptr;
    cx<T>* pp = nullptr;          // P': a1 itself (KB works in place) unless the scratch is interleaved (own buffer)
    cx<T>* slab = nullptr;
    cx<T>* tw = nullptr;
    cx<T>* phx = nullptr;
    cx<T>* phy = nullptr;
    cx<T>* phz = nullptr;
    T* dirw = nullptr;
    T* beta1 = nullptr;
    T* beta2 = nullptr;
    Segment* segs = nullptr;
    // exact-reduction mode only
    cx<T>* pseg = nullptr;        // [segment][x][y][z]
    Segment* segs_unit = nullptr; // one single-slot segment per pseg entry, same r
    T* ones = nullptr;
    cx<T>* rnyq = nullptr;        // Hermitian mode: Nyquist rows [slot][sign][kind][N/2-1][N]
    size_t slab_count = 0;
    // N = 16 whole-direction path: one partial Q_hat per workgroup
    T* small_part = nullptr;      // [small_wgs + 1][G] real partial results
    int small_wgs = 0, small_per = 0;

    template <typename U>
    bool dev_copy(U*& dst, const std::vector<U>& src) {
        const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(U);
        dst = (U*)be->alloc(bytes);
        if (!dst) return false;
        if (!src.empty()) be->upload(dst, src.data(), src.size() * sizeof(U));
        return true;
    }

    // A1' / A2' interleaved per element in one array (bfsm_core.hpp, ab_interleaved)
    bool pair_geometry() const { return plan.N == 128 && ab_interleaved<128, T>(); }
    bool interleaved() const { return pair_geometry() && !plan.hermitian; }
    int a_planes = 0;        // lx planes of A1' / A2' kept per direction (N, or N/2 + 1 in the Hermitian mode)
    size_t r_per_dir() const { return (size_t)4 * (plan.N / 2 - 1) * plan.N; }
    int max_batch = 1;       // distributions evaluated per call; scratch scales with it
    size_t cap = 1;          // directions resident at once (largest chunk)

    int init(const bfsm_desc& d, Backend* backend, std::string& err) {
        be = backend;
        plan = make_plan(d);
        HostTables<T> t = build_tables<T>(d, plan);
        const size_t G = plan.G();
        cap = (size_t)(plan.largest_chunk > 0 ? plan.largest_chunk : 1);
        max_batch = d.max_batch > 1 ? d.max_batch : 1;
        const size_t nb = (size_t)max_batch;
        slab_count = plan.segs.size();
        const size_t nslab = slab_count ? slab_count : 1;
        bool ok = true;
        ok = ok && (fhat = (cx<T>*)be->alloc(nb * G * sizeof(cx<T>)));
        ok = ok && (tg = (cx<T>*)be->alloc(nb * G * sizeof(cx<T>)));
        ok = ok && (tl = (cx<T>*)be->alloc(nb * G * sizeof(cx<T>)));
        ok = ok && (qhat = (cx<T>*)be->alloc(nb * G * sizeof(cx<T>)));
        a_planes = plan.hermitian ? plan.N / 2 + 1 : plan.N;
        const size_t Gp = (size_t)a_planes * plan.N * plan.N;       // elements of A1' / A2' per direction
        if (interleaved()) {
            // {A1', A2'} side by side in a1; the faithful mode's KB writes P' to a buffer of its own (see ab_interleaved)
            ok = ok && (a1 = (cx<T>*)be->alloc(2 * nb * cap * Gp * sizeof(cx<T>)));
            if (!plan.exact_reductions) ok = ok && (pp = (cx<T>*)be->alloc(nb * cap * G * sizeof(cx<T>)));
        } else {
            ok = ok && (a1 = (cx<T>*)be->alloc(nb * cap * Gp * sizeof(cx<T>)));
            ok = ok && (a2 = (cx<T>*)be->alloc(nb * cap * Gp * sizeof(cx<T>)));
            pp = a1;
        }
        if (plan.hermitian) ok = ok && (rnyq = (cx<T>*)be->alloc(nb * cap * r_per_dir() * sizeof(cx<T>)));
        ok = ok && (slab = (cx<T>*)be->alloc(nb * nslab * G * sizeof(cx<T>)));
        ok = ok && dev_copy(tw, t.tw) && dev_copy(phx, t.phx) && dev_copy(phy, t.phy) && dev_copy(phz, t.phz);
        ok = ok && dev_copy(dirw, t.dirw) && dev_copy(beta1, t.beta1) && dev_copy(beta2, t.beta2) && dev_copy(segs, plan.segs);
        if (ok && plan.exact_reductions) {
            std::vector<Segment> unit(plan.segs.size());
            for (size_t i = 0; i < unit.size(); ++i) unit[i] = Segment{(int)i, 1, plan.segs[i].r, 0};
            std::vector<T> one(unit.size() ? unit.size() : 1, (T)1);
            ok = ok && (pseg = (cx<T>*)be->alloc(nb * nslab * G * sizeof(cx<T>)));
            ok = ok && dev_copy(segs_unit, unit) && dev_copy(ones, one);
        }
        if (ok && plan.N == SMALL_N && !(d.flags & BFSM_FLAG_NO_SMALL_PATH) && plan.n_dirs() > 0 && max_batch == 1) {
            // a whole direction fits one workgroup: single evaluations take the three-launch path of collide_small
            // (handles created for batches keep one set of kernels: a single evaluation on them is bitwise a batch member)
            const long long nd = plan.n_dirs();
            const long long wg = nd < 256 ? nd : 256;
            small_per = (int)((nd + wg - 1) / wg);
            small_wgs = (int)((nd + small_per - 1) / small_per);
            ok = ok && (small_part = (T*)be->alloc((size_t)(small_wgs + 1) * G * sizeof(T)));
        }
        if (!ok) { err = "device allocation failed"; return BFSM_ERR_NOMEM; }
        return BFSM_OK;
    }

    // Whole evaluation on the N = 16 path (two launches, see bfsm_core.hpp); Q = gain [- loss]
    bool small_path(int nb) const { return small_part != nullptr && nb == 1; }
    void collide_small(double* Q_dev, const double* f_dev, bool with_loss) {
        const double Gc = (double)plan.G() * cbytes();
        // the loss term: one more workgroup while the gain workgroups leave a CU free (one workgroup per CU: 136 KiB of
        // LDS), otherwise workgroup 0 carries it next to its own share
        const bool extra = with_loss && small_wgs < 256;
        SmallGainParams<T> kg{f_dev, small_part, beta2, with_loss ? (extra ? small_wgs : 0) : -1, small_wgs, phx, phy, phz, dirw,
                              beta1, plan.n2stride, (int)plan.n_dirs(), small_per, plan.dir_begin, plan.sph_eff,
                              plan.exact_reductions ? 1 : 0};
        const int n_part = small_wgs + (extra ? 1 : 0);
        be->mark(BFSM_K_GAIN_LINE, 0.5 * n_part * Gc);
        be->template launch_small<SK::Gain, T>(n_part, kg);
        SmallReduceParams<T> kr{small_part, Q_dev, n_part};
        be->mark(BFSM_K_REDUCE, 0.5 * (n_part + 1.0) * Gc);
        be->template launch_small<SK::Reduce, T>((int)(plan.G() / 16), kr);
    }

    void destroy() {
        if (!be) return;
        if (small_part) { be->release(small_part); small_part = nullptr; }
        if (pp && pp != a1) be->release(pp);
        pp = nullptr;
        void* ptrs[] = {fhat, tg, tl, qhat, a1, a2, slab, tw, phx, phy, phz, dirw, beta1, beta2, segs, pseg, segs_unit, ones, rnyq};
        for (void* p : ptrs) if (p) be->release(p);
        fhat = tg = tl = qhat = a1 = a2 = slab = tw = phx = phy = phz = nullptr;
        dirw = beta1 = beta2 = nullptr;
        segs = nullptr;
        pseg = nullptr; segs_unit = nullptr; ones = nullptr; rnyq = nullptr;
    }

    double cbytes() const { return (double)sizeof(cx<T>); }
    int line_blocks() const { return plan.N * plan.N / line_npl(plan.N); }   // Wg<N>::NPL columns per workgroup
    static bool kn_rides_along(int n) { return n == 64 && nyq_rides_along<64>(); }

    // f_hat = FFT(f), then the gain term of this shard into qhat (partial Q_gain_hat, spectral layout).
    // nb distributions f_dev[nb][G] are processed by the same launches (grid.z / grid.y = batch member); all
    // per-distribution buffers are [nb][...].
    // Fusing the slab reduce into the first tail kernel saves a launch and a pass over Q_hat, but that kernel has only
    // N (x 2) workgroups, each walking all slabs: it pays for few slabs (a 1/8 shard of cfg3: 0.445 -> 0.439 ms per
    // evaluation) and loses against the wide reduce kernel for many (cfg2, 64 slabs: 4.0 k -> 2.8 k evaluations/s).
    bool fuse_reduce() const { return slab_count <= 8; }
    bool batch_together() const { return true; }          // batches of distributions share every launch

    // do_reduce = false leaves the slabs un-summed (qhat is not written): for callers that continue with
    // finish(..., from_slabs = true) in the same call and never expose qhat.
    void gain_partial(const double* f_dev, int nb = 1, bool do_reduce = true) {
        const int N = plan.N;
        const double Gc = (double)plan.G() * cbytes() * nb;
        const size_t G = plan.G();
        const size_t a_bs = cap * (size_t)a_planes * N * N, s_bs = (slab_count ? slab_count : 1) * G;
        const size_t r_bs = cap * r_per_dir();
        const double hfrac = (double)a_planes / N;                 // share of A' actually stored
        {   // F1: f_hat  (CUDABoltzmannOperator.cu:133-140)
            TileFwdRealParams<T> pa{f_dev, tg, tw};
            be->mark(BFSM_K_FFT_F, 1.5 * Gc);
            be->template launch<K::TileFwdReal, T>(N, nb, 1, pa, N);
            LineParams<T> pb{tg, fhat, tw};
            be->mark(BFSM_K_FFT_F, 2.0 * Gc);
            be->template launch<K::LineFwd, T>(line_blocks(), nb, 1, pb, N);
        }
        for (const Chunk& c : plan.chunks) {
            // KA's parallelism is planes x direction groups: keep >= 2 workgroups per CU when only N/2 + 1 planes run
            const int tw_ = target_workgroups(N, max_batch);
            // one resident wave of workgroups, except at N = 64: two rounds of half-length workgroups let the CUs that
            // finish early take more of them (measured, same box, alternating builds: KA 3.25 - 3.28 -> 3.17 - 3.20 ms at
            // config 4, 0.99 - 1.00 -> 0.98 ms at config 3, no change at N = 128: profiles/r03_ka_rounds_ab.txt)
#ifdef BFSM_KA_GROUPS_MULT          // A/B builds (tools only): rounds of KA workgroups
            const int ka_rounds = (BFSM_KA_GROUPS_MULT);
#else
            const int ka_rounds = N == 64 ? 2 : 1;
#endif
            const int groups_a = (tw_ / a_planes > 0 ? tw_ / a_planes : 1) * ka_rounds;
            const int per_group_a = (c.n + groups_a - 1) / groups_a;
            // phase tables beyond ~3 MiB do not stay in an XCD's L2 next to the streams: KA touches the rows ahead
            // (N = 64, config 4: 3 x 2.4 MiB of tables, KA 3.88 -> 3.66 ms; N = 128, config 5: 62 -> 52 ms)
            const bool warm = N >= 64 && 3.0 * (double)plan.n_dirs() * N * sizeof(cx<T>) > 3.0 * 1024 * 1024;
            const int ga = (c.n + per_group_a - 1) / per_group_a;
            // Hermitian mode, N = 64: the chunk's 2 c.n Nyquist-row workgroups (KN) ride along as extra rows of KA's grid
            const bool kn_rides = plan.hermitian && kn_rides_along(N);
            const int kn_blocks = kn_rides ? 2 * c.n : 0, kn_rows = (kn_blocks + a_planes - 1) / a_planes;
            GainInvParams<T> ka{fhat, a1, a2, phx, phy, phz, tw, c.dir0, c.n, per_group_a, a_bs, a_planes, warm ? 1 : 0};
            be->mark(BFSM_K_GAIN_INV, 2.0 * c.n * Gc * hfrac);
            if (kn_rides) {                     // KA + guest KN workgroups
                GainInvNyqParams<T> kan{ka, rnyq, r_bs, kn_blocks, ga};
                be->template launch<K::GainInvNyq, T>(a_planes, ga + kn_rows, nb, kan, N);
            } else if (pair_geometry() && !interleaved()) be->template launch<K::GainInvTwo, T>(a_planes, ga, nb, ka, N);
            else be->template launch<K::GainInv, T>(a_planes, ga, nb, ka, N);
            if (!plan.exact_reductions) {
                const size_t p_bs = interleaved() ? cap * G : a_bs;
                GainLineParams<T> kb{a1, a2, tw, a_bs, pp, p_bs};
                be->mark(BFSM_K_GAIN_LINE, 3.0 * c.n * Gc);
                be->template launch<K::GainLine, T>(line_blocks(), c.n, nb, kb, N);
                GainFwdParams<T> kc{pp, slab, dirw, segs, tw, c.dir0, c.seg0, p_bs, s_bs};
                be->mark(BFSM_K_GAIN_FWD, 1.0 * c.n * Gc);
                be->template launch<K::GainFwd, T>(N, c.n_seg, nb, kc, N);
            } else if (!plan.hermitian) {
                GainLineAccParams<T> kb{a1, a2, pseg, dirw, segs, tw, c.dir0, c.seg0, a_bs, s_bs};
                be->mark(BFSM_K_GAIN_LINE, (2.0 * c.n + c.n_seg) * Gc);
                be->template launch<K::GainLineAcc, T>(line_blocks(), c.n_seg, nb, kb, N);
            } else {
                if (!kn_rides) {
                    NyqRowsParams<T> kn{fhat, rnyq, phx, phy, phz, tw, c.dir0, r_bs};
                    const int npl = line_npl(N), ncol = 2 * (N / 2 - 1);
                    be->mark(BFSM_K_GAIN_INV, 0.0);
                    be->template launch<K::NyqRows, T>((ncol + npl - 1) / npl, 2 * c.n, nb, kn, N);
                }
                GainLineAccHParams<T> kb{a1, a2, rnyq, pseg, dirw, segs, tw, c.dir0, c.seg0, a_bs, s_bs, r_bs};
                be->mark(BFSM_K_GAIN_LINE, (2.0 * c.n * hfrac + c.n_seg) * Gc);
                be->template launch<K::GainLineAccH, T>(line_blocks(), c.n_seg, nb, kb, N);
            }
        }
        if (plan.exact_reductions && slab_count) {   // one forward tile pass per segment, all chunks at once
            GainFwdParams<T> kc{pseg, slab, ones, segs_unit, tw, 0, 0, s_bs, s_bs};
            be->mark(BFSM_K_GAIN_FWD, 1.0 * (double)slab_count * Gc);
            be->template launch<K::GainFwd, T>(N, (int)slab_count, nb, kc, N);
        }
        if (do_reduce) {
            ReduceParams<T> kr{slab, qhat, beta1, segs, (int)slab_count, plan.n2stride, s_bs};
            be->mark(BFSM_K_REDUCE, ((double)slab_count + 1.0) * Gc);
            be->template launch<K::Reduce, T>((int)((plan.G() + 255) / 256), nb, 1, kr, N);
        }
    }

    // Loss term + final inverse transforms + combine  (CUDABoltzmannOperator.cu:193-216)
    // with_loss = false: Q = Re IFFT(qhat) only -- the partial result a rank contributes when the caller sums Q
    // itself (half the bytes of summing Q_hat) and another rank adds the loss term.
    // from_slabs = true: the slab reduce is fused into the first tail kernel (after gain_partial(.., false)).
    void finish(double* Q_dev, const double* f_dev, bool with_loss = true, int nb = 1, bool from_slabs = false) {
        const int N = plan.N;
        const double Gc = (double)plan.G() * cbytes() * nb;
        const size_t s_bs = (slab_count ? slab_count : 1) * plan.G();
        TailInvParams<T> ta{qhat, fhat, beta2, tg, (long long)(tl - tg), tw, slab, beta1, segs, from_slabs ? (int)slab_count : -1,
                            plan.n2stride, s_bs};
        be->mark(BFSM_K_TAIL, ((with_loss ? 4.0 : 2.0) + (from_slabs ? (double)slab_count - 1.0 : 0.0)) * Gc);
        be->template launch<K::TailInv, T>(N, with_loss ? 2 : 1, nb, ta, N);
        TailLineParams<T> tb{tg, tl, f_dev, Q_dev, tw, with_loss ? 1 : 0};
        be->mark(BFSM_K_TAIL, (with_loss ? 3.0 : 1.5) * Gc);
        be->template launch<K::TailLine, T>(line_blocks(), nb, 1, tb, N);
    }

    // In-place batched 3-D transform on user data (bfsm_fft3d)
    int fft3d(cx<T>* data, int batch, int sign) {
        const int N = plan.N;
        LineParams<T> p{data, data, tw};
        if (sign < 0) {
            be->mark(-1, 0);
            be->template launch<K::TileFwd, T>(N, batch, 1, p, N);
            be->mark(-1, 0);
            be->template launch<K::LineFwd, T>(line_blocks(), batch, 1, p, N);
        } else {
            be->mark(-1, 0);
            be->template launch<K::LineInv, T>(line_blocks(), batch, 1, p, N);
            be->mark(-1, 0);
            be->template launch<K::TileInv, T>(N, batch, 1, p, N);
        }
        return BFSM_OK;
    }
};

// SURVEY.md 8(d) model for this shard, always in FULL quadrature directions (3 FFTs per direction, read + write each)
inline double alg_bytes_per_eval(const PlanInfo& p) {
    const double c = p.precision == BFSM_F64 ? 16.0 : 8.0;
    return (6.0 * (double)(p.full_end - p.full_begin) + 9.0) * (double)p.G() * c;
}

// Bytes the launch sequence actually moves (model): equals the figure above in the faithful mode (+ slabs); in the
// exact-reduction mode 4 array passes per effective direction + 2 per segment.
inline double moved_bytes_per_eval(const PlanInfo& p) {
    const double c = p.precision == BFSM_F64 ? 16.0 : 8.0;
    const double G = (double)p.G(), n = (double)p.n_dirs(), sg = (double)p.segs.size();
    // size-generic path: one pass per axis (or x + a fused (y,z) plane pass), pointwise steps fused on the load side
    if (p.N == 0) return ((double)p.gen_moves * n + 2.0 * p.gen_slabs + (p.gen_plane ? 18.0 : 27.0)) * G * c;
    if (!p.exact_reductions) return (6.0 * n + 2.0 * sg + 9.0) * G * c;
    const double h = p.hermitian ? (double)(p.N / 2 + 1) / p.N : 1.0;
    return (4.0 * n * h + 4.0 * sg + 9.0) * G * c;
}

}  // namespace bfsm
