/*
 * bfsm_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity oracle).
 *
 * A plain-C, CPU restatement of the reference's Fourier-spectral Boltzmann collision operator
 * (FFTW backend).  It is the checker for the HIP path: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  Nothing under boltzmann-fourier-spectral-method_amd/
 * links, imports or calls it.
 *
 * Parity pinning: the reference itself cannot be built in this image (it needs FFTW3 and GSL, both
 * absent; writing stand-in headers for them is not allowed), so oracle/_ref does not exist.  The oracle
 * is pinned by the reference's own published known-answer values (Results/maxwell_bkw_fftw_atomics.txt,
 * BKW L1/L2/Linf to 9 digits; see tests/golden/bkw_norms.json and tests/test_oracle_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 *
 * Third-party arithmetic the reference delegates and that is restated here from the published algorithm:
 *   - FFTW3 (unpinned version): unnormalised c2c 3-D DFT, FORWARD = exp(-i...), BACKWARD = exp(+i...)
 *     (call sites Collisions/FFTWBoltzmannOperator.cpp:64-65,186,229-230,249,305,309)
 *   - GSL (unpinned): gsl_integration_glfixed_point -> Gauss-Legendre nodes ascending on [a,b],
 *     x = (a+b)/2 + (b-a)/2 t_i, w = (b-a)/2 w_i   (call site Quadratures/GaussLegendre.hpp:14-23)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_PI 3.14159265358979323846 /* Utilities/constants.hpp:7 */

typedef struct { double re, im; } cplx;

/* ------------------------------------------------------------------------------------------------
 * 1-D DFT plans.  Power-of-two lengths: iterative radix-2 with a long-double twiddle table.
 * Other lengths: direct O(n^2) DFT (the oracle only has to be right, not fast).
 * sign = -1: FFTW_FORWARD, sign = +1: FFTW_BACKWARD.  Unnormalised, like FFTW.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int n;
    int pow2;
    cplx* tw;   /* tw[k] = exp(-2 pi i k / n), k < n  (forward); conjugated on the fly for backward */
    int* brev;  /* bit reversal (pow2 only) */
} plan1d;

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

static plan1d* plan1d_create(int n) {
    plan1d* p = (plan1d*)calloc(1, sizeof(plan1d));
    p->n = n;
    p->pow2 = is_pow2(n);
    p->tw = (cplx*)malloc(sizeof(cplx) * (size_t)n);
    for (int k = 0; k < n; ++k) {
        long double a = -2.0L * 3.141592653589793238462643383279502884L * (long double)k / (long double)n;
        p->tw[k].re = (double)cosl(a);
        p->tw[k].im = (double)sinl(a);
    }
    if (p->pow2) {
        int lg = 0;
        while ((1 << lg) < n) ++lg;
        p->brev = (int*)malloc(sizeof(int) * (size_t)n);
        for (int i = 0; i < n; ++i) {
            int r = 0;
            for (int b = 0; b < lg; ++b) if (i & (1 << b)) r |= 1 << (lg - 1 - b);
            p->brev[i] = r;
        }
    }
    return p;
}

static void plan1d_destroy(plan1d* p) {
    if (!p) return;
    free(p->tw);
    free(p->brev);
    free(p);
}

/* In-place transform of one contiguous line x[0..n).  scratch must hold n elements. */
static void dft1d(const plan1d* p, cplx* x, cplx* scratch, int sign) {
    const int n = p->n;
    if (n == 1) return;
    if (p->pow2) {
        for (int i = 0; i < n; ++i) {
            int j = p->brev[i];
            if (j > i) { cplx t = x[i]; x[i] = x[j]; x[j] = t; }
        }
        for (int len = 2; len <= n; len <<= 1) {
            const int half = len >> 1, step = n / len;
            for (int base = 0; base < n; base += len) {
                for (int j = 0; j < half; ++j) {
                    const cplx w = p->tw[j * step];
                    const double wr = w.re, wi = (sign < 0) ? w.im : -w.im;
                    cplx* a = x + base + j;
                    cplx* b = a + half;
                    const double tr = b->re * wr - b->im * wi;
                    const double ti = b->re * wi + b->im * wr;
                    b->re = a->re - tr; b->im = a->im - ti;
                    a->re += tr;        a->im += ti;
                }
            }
        }
    } else {
        for (int k = 0; k < n; ++k) {
            long double sr = 0, si = 0;
            for (int j = 0; j < n; ++j) {
                const cplx w = p->tw[(int)(((long long)j * k) % n)];
                const double wr = w.re, wi = (sign < 0) ? w.im : -w.im;
                sr += (long double)x[j].re * wr - (long double)x[j].im * wi;
                si += (long double)x[j].re * wi + (long double)x[j].im * wr;
            }
            scratch[k].re = (double)sr; scratch[k].im = (double)si;
        }
        memcpy(x, scratch, sizeof(cplx) * (size_t)n);
    }
}

typedef struct {
    int nx, ny, nz;
    plan1d *px, *py, *pz;
} plan3d;

static plan3d* plan3d_create(int nx, int ny, int nz) {
    plan3d* p = (plan3d*)calloc(1, sizeof(plan3d));
    p->nx = nx; p->ny = ny; p->nz = nz;
    p->px = plan1d_create(nx);
    p->py = plan1d_create(ny);
    p->pz = plan1d_create(nz);
    return p;
}

static void plan3d_destroy(plan3d* p) {
    if (!p) return;
    plan1d_destroy(p->px); plan1d_destroy(p->py); plan1d_destroy(p->pz);
    free(p);
}

/* Restates fftw_execute_dft(plan_dft_3d) (FFTWBoltzmannOperator.cpp:64-65): row-major [nx][ny][nz],
 * nz contiguous, unnormalised, out-of-place allowed (in may equal out).  work >= 2*max(nx,ny,nz). */
static void dft3d(const plan3d* p, const cplx* in, cplx* out, int sign, cplx* work) {
    const int nx = p->nx, ny = p->ny, nz = p->nz;
    const size_t g = (size_t)nx * ny * nz;
    if (in != out) memcpy(out, in, sizeof(cplx) * g);
    int nmax = nx > ny ? nx : ny; if (nz > nmax) nmax = nz;
    cplx* line = work;
    cplx* scratch = work + nmax;
    /* z lines (contiguous) */
    for (size_t l = 0; l < (size_t)nx * ny; ++l) dft1d(p->pz, out + l * nz, scratch, sign);
    /* y lines (stride nz) */
    for (int i = 0; i < nx; ++i)
        for (int k = 0; k < nz; ++k) {
            cplx* base = out + (size_t)i * ny * nz + k;
            for (int j = 0; j < ny; ++j) line[j] = base[(size_t)j * nz];
            dft1d(p->py, line, scratch, sign);
            for (int j = 0; j < ny; ++j) base[(size_t)j * nz] = line[j];
        }
    /* x lines (stride ny*nz) */
    for (int j = 0; j < ny; ++j)
        for (int k = 0; k < nz; ++k) {
            cplx* base = out + (size_t)j * nz + k;
            for (int i = 0; i < nx; ++i) line[i] = base[(size_t)i * ny * nz];
            dft1d(p->px, line, scratch, sign);
            for (int i = 0; i < nx; ++i) base[(size_t)i * ny * nz] = line[i];
        }
}

/* Exposed for the FFT unit tests (mirrors the round-trip check of fftw_benchmark.cpp:137-171).
 * data: interleaved (re,im) doubles, [nx][ny][nz]; sign -1 forward / +1 backward; in place. */
int bfsm_oracle_fft3d(int nx, int ny, int nz, double* data, int sign) {
    if (nx < 1 || ny < 1 || nz < 1 || (sign != 1 && sign != -1)) return 1;
    plan3d* p = plan3d_create(nx, ny, nz);
    int nmax = nx > ny ? nx : ny; if (nz > nmax) nmax = nz;
    cplx* work = (cplx*)malloc(sizeof(cplx) * 2 * (size_t)nmax);
    dft3d(p, (const cplx*)data, (cplx*)data, sign, work);
    free(work);
    plan3d_destroy(p);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Gauss-Legendre on [a,b], ascending nodes -- what Quadratures/GaussLegendre.hpp:10-24 obtains from
 * gsl_integration_glfixed_point(a, b, i, &x, &w, table): x_i = (a+b)/2 + (b-a)/2 t_i,
 * w_i = (b-a)/2 w_i, t_i the i-th root of P_n in increasing order.  Newton on P_n in long double.
 * ---------------------------------------------------------------------------------------------- */
int bfsm_oracle_gauss_legendre(int n, double a, double b, double* nodes, double* weights) {
    if (n < 1) return 1;
    const long double PI_L = 3.141592653589793238462643383279502884L;
    const long double half = ((long double)b - (long double)a) / 2, mid = ((long double)a + (long double)b) / 2;
    for (int i = 0; i < (n + 1) / 2; ++i) {
        long double x = cosl(PI_L * ((long double)i + 0.75L) / ((long double)n + 0.5L)); /* descending roots */
        long double dp = 1;
        for (int it = 0; it < 100; ++it) {
            long double p0 = 1, p1 = x;
            for (int k = 2; k <= n; ++k) {
                long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
                p0 = p1; p1 = pk;
            }
            if (n == 1) { p0 = 1; p1 = x; }
            dp = n * (x * p1 - p0) / (x * x - 1);
            long double dx = p1 / dp;
            x -= dx;
            if (fabsl(dx) < 1e-19L) {
                /* refresh derivative at the converged root */
                p0 = 1; p1 = x;
                for (int k = 2; k <= n; ++k) {
                    long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
                    p0 = p1; p1 = pk;
                }
                dp = n * (x * p1 - p0) / (x * x - 1);
                break;
            }
        }
        long double w = 2 / ((1 - x * x) * dp * dp);
        /* x is the i-th largest root; ascending order puts it at n-1-i and -x at i */
        nodes[n - 1 - i] = (double)(mid + half * x);
        nodes[i] = (double)(mid - half * x);
        weights[n - 1 - i] = (double)(half * w);
        weights[i] = (double)(half * w);
    }
    if (n % 2 == 1) nodes[n / 2] = (double)mid; /* exact centre root */
    return 0;
}

/* sincc(x) = sin(x+eps)/(x+eps), eps = DBL_EPSILON  (Collisions/FFTWBoltzmannOperator.hpp:17-21) */
static double sincc(double x) {
    const double eps = DBL_EPSILON;
    return sin(x + eps) / (x + eps);
}

/* Fourier modes in FFT order 0..n/2-1, -n/2..-1  (FFTWBoltzmannOperator.cpp:50-57) */
static void fill_modes(int n, int* l) {
    int c = 0;
    for (int i = 0; i < n / 2; ++i) l[c++] = i;
    for (int i = -n / 2; i < 0; ++i) l[c++] = i;
}

typedef struct {
    int nvx, nvy, nvz;
    int n_gl, n_sph;
    const double* gl_nodes;
    const double* gl_wts;
    const double* sph_wts;
    const double* sx;
    const double* sy;
    const double* sz;
    double gamma, b_gamma, L;
} bfsm_oracle_desc;

/*
 * Restates BoltzmannOperator<FFTW_Backend>::computeCollision (Collisions/FFTWBoltzmannOperator.cpp:147-334)
 * step for step.  Differences from the reference that do not change the mathematics:
 *   - directions are streamed (per-thread scratch of one direction) instead of materialising 6 arrays of
 *     B*G complex (cpp:30-37), so configs 4/5 fit in memory and 64-bit offsets are used throughout;
 *   - the two `omp atomic` updates into Q_gain_hat (cpp:267-270) become thread-private accumulators summed
 *     in thread order (deterministic; the reference's own run-to-run spread from atomics is ~1e-16 rel);
 *   - dir_begin/dir_end restrict the (r,s) loop to flattened directions b = r*n_sph + s in [begin,end)
 *     (used to check the sharded multi-GPU path and for the bounded cpu_baseline sample);
 *     the loss term is always computed in full.  Pass 0, n_gl*n_sph for the reference behaviour.
 * f_in, Q: real [nvx][nvy][nvz].  qhat_out (optional): the (partial) Q_gain_hat before the final inverse
 * FFT, interleaved complex [nvx][nvy][nvz] -- what one GPU shard contributes to the reduce.
 * Returns 0 on success.
 */
int bfsm_oracle_collide_ex(const bfsm_oracle_desc* d, const double* f_in, double* Q, double* qhat_out,
                           long long dir_begin, long long dir_end, int n_threads) {
    const int Nvx = d->nvx, Nvy = d->nvy, Nvz = d->nvz;
    if (Nvx < 2 || Nvy < 2 || Nvz < 2 || (Nvx | Nvy | Nvz) & 1) return 1; /* mode tables need even sizes */
    const int N_gl = d->n_gl, N_sph = d->n_sph;
    const long long B = (long long)N_gl * N_sph;
    if (dir_begin < 0 || dir_end > B || dir_begin > dir_end) return 2;
    const size_t G = (size_t)Nvx * Nvy * Nvz;
    const double fft_scale = 1.0 / (double)G;          /* cpp:162 */
    const double pi = ORACLE_PI, L = d->L, gamma = d->gamma, b_gamma = d->b_gamma;

    int* lx = (int*)malloc(sizeof(int) * Nvx);
    int* ly = (int*)malloc(sizeof(int) * Nvy);
    int* lz = (int*)malloc(sizeof(int) * Nvz);
    fill_modes(Nvx, lx); fill_modes(Nvy, ly); fill_modes(Nvz, lz);

    plan3d* plan = plan3d_create(Nvx, Nvy, Nvz);
    int nmax = Nvx > Nvy ? Nvx : Nvy; if (Nvz > nmax) nmax = Nvz;

    cplx* f = (cplx*)malloc(sizeof(cplx) * G);
    cplx* f_hat = (cplx*)malloc(sizeof(cplx) * G);
    cplx* Q_gain_hat = (cplx*)calloc(G, sizeof(cplx));
    cplx* work0 = (cplx*)malloc(sizeof(cplx) * 2 * (size_t)nmax);

    /* cpp:168-180: promote f to complex, zero Q_gain_hat */
    for (size_t i = 0; i < G; ++i) { f[i].re = f_in[i]; f[i].im = 0.0; }
    /* cpp:185-186: f_hat = fft(f) */
    dft3d(plan, f, f_hat, -1, work0);

#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    cplx** acc = (cplx**)calloc((size_t)n_threads, sizeof(cplx*));
    int alloc_fail = 0;

#pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        cplx* a1 = (cplx*)malloc(sizeof(cplx) * G);   /* alpha1_times_f(_hat) for one direction */
        cplx* a2 = (cplx*)malloc(sizeof(cplx) * G);   /* alpha2_times_f(_hat) */
        cplx* my = (cplx*)calloc(G, sizeof(cplx));    /* private Q_gain_hat partial */
        cplx* work = (cplx*)malloc(sizeof(cplx) * 2 * (size_t)nmax);
        acc[tid] = my;
        if (!a1 || !a2 || !my || !work) {
#pragma omp atomic write
            alloc_fail = 1;
        }
#pragma omp barrier
        if (!alloc_fail) {
            /* cpp:191-193: loop over the batches (directions) in parallel */
#pragma omp for schedule(dynamic, 1)
            for (long long b = dir_begin; b < dir_end; ++b) {
                const int r = (int)(b / N_sph), s = (int)(b % N_sph);   /* b = r*N_spherical + s, cpp:196 */
                /* cpp:198-225: phase multiply */
                for (int i = 0; i < Nvx; ++i)
                    for (int j = 0; j < Nvy; ++j)
                        for (int k = 0; k < Nvz; ++k) {
                            const size_t idx3 = ((size_t)i * Nvy + j) * Nvz + k;
                            const double l_dot_sigma = lx[i] * d->sx[s] + ly[j] * d->sy[s] + lz[k] * d->sz[s];
                            const double tmp = -(pi / (2 * L)) * d->gl_nodes[r] * l_dot_sigma;
                            const double a_re = cos(tmp), a_im = sin(tmp);
                            const double b_re = f_hat[idx3].re, b_im = f_hat[idx3].im;
                            a1[idx3].re = fft_scale * (a_re * b_re - a_im * b_im);
                            a1[idx3].im = fft_scale * (a_re * b_im + a_im * b_re);
                            a2[idx3].re = fft_scale * (a_re * b_re + a_im * b_im);
                            a2[idx3].im = fft_scale * (a_re * b_im - a_im * b_re);
                        }
                /* cpp:229-230: two inverse FFTs */
                dft3d(plan, a1, a1, +1, work);
                dft3d(plan, a2, a2, +1, work);
                /* cpp:233-246: product (full complex multiply, not conj) */
                for (size_t i = 0; i < G; ++i) {
                    const double a_re = a1[i].re, a_im = a1[i].im, b_re = a2[i].re, b_im = a2[i].im;
                    a1[i].re = a_re * b_re - (a_im * b_im);
                    a1[i].im = a_re * b_im + (a_im * b_re);
                }
                /* cpp:249: forward FFT of the product */
                dft3d(plan, a1, a1, -1, work);
                /* cpp:252: weight */
                const double weight = fft_scale * d->gl_wts[r] * d->sph_wts[s] * pow(d->gl_nodes[r], gamma + 2);
                /* cpp:254-273: beta1 and accumulation */
                for (int i = 0; i < Nvx; ++i)
                    for (int j = 0; j < Nvy; ++j)
                        for (int k = 0; k < Nvz; ++k) {
                            const size_t idx3 = ((size_t)i * Nvy + j) * Nvz + k;
                            const double norm_l = sqrt((double)(lx[i] * lx[i] + ly[j] * ly[j] + lz[k] * lz[k]));
                            const double beta1 = 4 * pi * b_gamma * sincc(pi * d->gl_nodes[r] * norm_l / (2 * L));
                            my[idx3].re += weight * beta1 * a1[idx3].re;
                            my[idx3].im += weight * beta1 * a1[idx3].im;
                        }
            }
        }
        free(a1); free(a2); free(work);
    }
    if (alloc_fail) {
        for (int t = 0; t < n_threads; ++t) free(acc[t]);
        free(acc); free(f); free(f_hat); free(Q_gain_hat); free(work0);
        free(lx); free(ly); free(lz); plan3d_destroy(plan);
        return 3;
    }
    for (int t = 0; t < n_threads; ++t) {
        if (!acc[t]) continue;
        for (size_t i = 0; i < G; ++i) { Q_gain_hat[i].re += acc[t][i].re; Q_gain_hat[i].im += acc[t][i].im; }
        free(acc[t]);
    }
    free(acc);

    if (qhat_out) memcpy(qhat_out, Q_gain_hat, sizeof(cplx) * G);

    /* cpp:281-299: beta2 * f_hat */
    cplx* b2f = (cplx*)malloc(sizeof(cplx) * G);
    for (int i = 0; i < Nvx; ++i)
        for (int j = 0; j < Nvy; ++j)
            for (int k = 0; k < Nvz; ++k) {
                const size_t idx3 = ((size_t)i * Nvy + j) * Nvz + k;
                double beta2 = 0.0;
                const double norm_l = sqrt((double)(lx[i] * lx[i] + ly[j] * ly[j] + lz[k] * lz[k]));
                for (int r = 0; r < N_gl; ++r)
                    beta2 += 16 * pi * pi * b_gamma * d->gl_wts[r] * pow(d->gl_nodes[r], gamma + 2) *
                             sincc(pi * d->gl_nodes[r] * norm_l / L);
                b2f[idx3].re = fft_scale * beta2 * f_hat[idx3].re;
                b2f[idx3].im = fft_scale * beta2 * f_hat[idx3].im;
            }
    /* cpp:304-309: Q_gain = ifft(Q_gain_hat), beta2_times_f = ifft(beta2_times_f_hat) */
    dft3d(plan, Q_gain_hat, Q_gain_hat, +1, work0);
    dft3d(plan, b2f, b2f, +1, work0);
    /* cpp:314-330: Q = Re(Q_gain) - Re(beta2_times_f * f) */
    for (size_t i = 0; i < G; ++i) {
        const double a_re = b2f[i].re, a_im = b2f[i].im, b_re = f[i].re, b_im = f[i].im;
        const double loss_re = a_re * b_re - (a_im * b_im);
        Q[i] = Q_gain_hat[i].re - loss_re;
    }

    free(b2f); free(f); free(f_hat); free(Q_gain_hat); free(work0);
    free(lx); free(ly); free(lz);
    plan3d_destroy(plan);
    return 0;
}

int bfsm_oracle_collide(const bfsm_oracle_desc* d, const double* f_in, double* Q,
                        long long dir_begin, long long dir_end, int n_threads) {
    return bfsm_oracle_collide_ex(d, f_in, Q, 0, dir_begin, dir_end, n_threads);
}

/* Number of OpenMP threads the oracle would use by default (reported as cpu_baseline.cores). */
int bfsm_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* BKW (Bobylev-Krook-Wu) known-answer pair on the reference driver's grid
 * (maxwell_bkw_fftw.cpp:54-99): gamma=0, b_gamma=1/(4pi), S=5, R=2S, L=(3+sqrt2)/2*S, t=6.5.
 * Fills f and Q_exact on the N^3 grid v_i = -L + dv/2 + i dv and returns dv through *dv_out. */
void bfsm_oracle_bkw(int Nv, double S, double t, double* f, double* Q_exact, double* L_out, double* dv_out) {
    const double pi = ORACLE_PI;
    const double L = ((3 + sqrt(2.0)) / 2) * S;
    const double dv = 2 * L / Nv;
    const double K = 1 - exp(-t / 6);
    const double dK = exp(-t / 6) / 6;
    for (int i = 0; i < Nv; ++i) {
        const double vx = -L + dv / 2 + i * dv;
        for (int j = 0; j < Nv; ++j) {
            const double vy = -L + dv / 2 + j * dv;
            for (int k = 0; k < Nv; ++k) {
                const double vz = -L + dv / 2 + k * dv;
                const size_t idx3 = ((size_t)i * Nv + j) * Nv + k;
                const double r_sq = vx * vx + vy * vy + vz * vz;
                double fv = exp(-(r_sq) / (2 * K)) * ((5 * K - 3) / K + (1 - K) / (pow(K, 2)) * (r_sq));
                fv *= 1 / (2 * pow(2 * pi * K, 1.5));
                double q = (-3 / (2 * K) + r_sq / (2 * pow(K, 2))) * fv;
                q += 1 / (2 * pow(2 * pi * K, 1.5)) * exp(-r_sq / (2 * K)) * (3 / (pow(K, 2)) + (K - 2) / (pow(K, 3)) * r_sq);
                q *= dK;
                f[idx3] = fv;
                Q_exact[idx3] = q;
            }
        }
    }
    if (L_out) *L_out = L;
    if (dv_out) *dv_out = dv;
}
