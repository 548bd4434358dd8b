/*
 * bfsm.h -- C-ABI of the MI355X-native Fourier-spectral Boltzmann collision operator (libbfsm_hip.so).
 *
 * This is the drop-in boundary for ONE path of i3s93/Boltzmann-Fourier-Spectral-Method: a single evaluation of
 * Q = Q(f,f) behind AbstractCollisionOperator::computeCollision.  Plain pointers and sizes only; no C++ or torch
 * types.  Each entry point names the reference interface it replaces (file:line relative to the reference root).
 * The C++ mirror of the reference's operator class (BoltzmannOperator<HIP_Backend>) and the Python ctypes binding
 * are thin wrappers over exactly these symbols; see INTEGRATION.md.
 *
 * Conventions (identical to the reference's CUDA backend, Collisions/CUDABoltzmannOperator.cu:119-220):
 *   - f and Q are DEVICE pointers to Nvx*Nvy*Nvz contiguous doubles, row-major [i][j][k], k (v_z) contiguous
 *     (maxwell_bkw_cuda.cu:119-126); they stay owned by the caller, all scratch is owned by the handle.
 *   - a handle is not re-entrant (shared scratch), like the reference object: one thread at a time per handle, and
 *     consecutive calls on one handle must be ordered on the device (same stream, or the caller's own events) because
 *     they reuse the scratch.  Different handles are independent and may be driven from different threads / streams.
 *   - every call makes the handle's device current for its own duration and restores the calling thread's current
 *     device before it returns.
 *   - the *_async entry points only enqueue kernels on the given stream: no allocation and no synchronisation after the
 *     first evaluation; per call one host-side capture-status query (hipStreamIsCapturing) and, outside a capture, one
 *     record of a handle-owned event behind the call's work (what bfsm_synchronize waits for; the event of a stream is
 *     created at its first use).  After one evaluation outside a capture they can therefore be captured into a HIP graph;
 *     nothing is recorded during a capture, so replays of such a graph are the caller's to synchronise.  The caller's
 *     stream handle is never used after the call returns: the caller may destroy the stream at any time (a NEW stream
 *     that reuses the handle value of a destroyed one takes over its entry: bfsm_synchronize then covers the new
 *     stream's work; the destroyed stream's work completes under hipStreamDestroy's own rules).
 *   - functions never throw and never exit: they return BFSM_OK or an error code, and bfsm_last_error() returns a
 *     human-readable message (the reference prints and std::exit()s, CUDABoltzmannOperator.hpp:20-38; the C++
 *     wrapper restores that behaviour).
 */
#ifndef BFSM_H
#define BFSM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFSM_VERSION 2   /* 2: any-box grids, bfsm_collide_batch_partial_async, BFSM_FLAG_NO_SMALL_PATH */

enum {
    BFSM_OK = 0,
    BFSM_ERR_INVALID = 1,      /* bad argument / descriptor */
    BFSM_ERR_UNSUPPORTED = 2,  /* grid size / precision combination without a kernel */
    BFSM_ERR_HIP = 3,          /* a HIP runtime call failed (message has file:line) */
    BFSM_ERR_NOMEM = 4
};

enum {
    BFSM_F64 = 64, /* IEEE double throughout (reference behaviour) */
    BFSM_F32 = 32  /* single-precision transforms and tables; f and Q stay double at the boundary */
};

enum {
    BFSM_FLAG_NONE = 0,
    BFSM_FLAG_PROFILE = 1, /* record HIP events around every kernel launch (bfsm_get_counters) */
    /* Opt-in exact work reductions; results equal the faithful path up to rounding order (~1e-16):
     *  (i)  antipodal pairs: if the spherical rule satisfies sigma_{s+M/2} == -sigma_s with equal weights (all shipped
     *       symmetric designs do, bit-exactly), direction s+M/2 gives the same product A1*A2 as s, so only M/2
     *       directions per radial node are evaluated, with doubled weight; otherwise this part is skipped;
     *  (ii) FFT linearity: the products of all directions of a radial node are summed in physical space and
     *       forward-transformed once, instead of one forward FFT per direction.
     * The default (flag clear) evaluates every direction: its own two inverse 3-D transforms, its own product and the
     * x part of its own forward transform; the (y,z) part of the forward transform is applied once to the weighted sum
     * of a run of directions that share a radial node (linearity; every array is still written and read once per
     * direction: 6 array passes per direction).  Build with -DBFSM_KC_PER_DIRECTION for one (y,z)
     * forward transform per direction (the reference's literal loop, FFTWBoltzmannOperator.cpp:249).
     * On boxes served by the size-generic path (see bfsm_desc::nvx) this flag and BFSM_FLAG_HERMITIAN are accepted and
     * have no effect (the results are the same by definition; bfsm_counters::exact_reductions reports 0). */
    BFSM_FLAG_EXACT_REDUCTIONS = 2,
    /* Additional exact reduction on top of BFSM_FLAG_EXACT_REDUCTIONS (invalid without it): f is real, so the
     * half-transformed arrays satisfy A'[-lx] = conj A'[lx] up to the three Nyquist planes; only the planes
     * lx = 0 .. N/2 are computed and stored, the rest is rebuilt by conjugation plus exact rank-one Nyquist terms
     * (the r2c / c2r saving the reference lists as future work, CUDABoltzmannOperator.cu:36). */
    BFSM_FLAG_HERMITIAN = 4,
    /* N = 16 only: do not use the whole-direction kernels (a direction kept in one workgroup's LDS, two launches
     * per evaluation) for single evaluations; the plane-tile pipeline of the larger grids is used instead.  Same
     * results up to rounding order; for comparisons and tests. */
    BFSM_FLAG_NO_SMALL_PATH = 8
};

typedef struct bfsm_plan* bfsm_handle;

/*
 * Operator description == the constructor arguments of BoltzmannOperator<CUDA_Backend>
 * (Collisions/CUDABoltzmannOperator.hpp:48-54) with the quadrature objects flattened to the arrays the operator
 * reads from them (getNodes/getWeights, Quadratures/AbstractQuadrature.hpp:17-29; getx/gety/getz/getWeights,
 * Quadratures/AbstractSphericalQuadratures.hpp:21-42).  All arrays are HOST pointers, copied during create.
 */
typedef struct bfsm_desc {
    int nvx, nvy, nvz;        /* velocity grid: every extent even, in [4, 256], prime factors 2, 3, 5, 7, 11, 13 (the
                                 reference plans any Nvx x Nvy x Nvz, CUDABoltzmannOperator.cu:86-100).  Cubes of 16, 24, 32,
                                 40, 48, 64, 80, 96, 128 run on the fused pipeline (6 array passes per direction); every other box
                                 on the size-generic path (run-time sizes: the same three fused kernels where a (y,z) plane
                                 fits the LDS, per-axis passes otherwise; a half to a sixth of the fused pipeline's speed),
                                 both precisions.  Anything else: BFSM_ERR_UNSUPPORTED */
    int n_gl;                 /* Gauss-Legendre points (radial)          */
    int n_sph;                /* spherical quadrature points             */
    const double* gl_nodes;   /* [n_gl]  rho_r on [0,R]                   */
    const double* gl_wts;     /* [n_gl]                                   */
    const double* sph_wts;    /* [n_sph]                                  */
    const double* sx;         /* [n_sph] unit vectors                     */
    const double* sy;
    const double* sz;
    double gamma;             /* kernel exponent (0: Maxwell molecules)   */
    double b_gamma;           /* kernel constant                          */
    double L;                 /* half-width of the periodic velocity box  */
    int precision;            /* BFSM_F64 | BFSM_F32                      */
    int device;               /* HIP device ordinal                       */
    long long dir_begin;      /* shard of the flattened quadrature directions b = r*n_sph + s handled by this   */
    long long dir_end;        /* handle: [dir_begin, dir_end).  0,0 = all directions (single-GPU behaviour).     */
    int max_chunk;            /* directions resident at once (0 = default 1024: the whole shard in one pass when it
                                 fits; scratch = 2 * chunk * G complex); bounds scratch, not results                 */
    int flags;                /* BFSM_FLAG_*                              */
    int max_batch;            /* distributions per bfsm_collide_batch call (0/1 = single); scratch scales with it */
} bfsm_desc;

/* Per-kernel accounting filled when BFSM_FLAG_PROFILE is set (all zero otherwise). */
enum { BFSM_K_FFT_F = 0, BFSM_K_GAIN_INV = 1, BFSM_K_GAIN_LINE = 2, BFSM_K_GAIN_FWD = 3, BFSM_K_REDUCE = 4,
       BFSM_K_TAIL = 5, BFSM_K_COUNT = 6 };
typedef struct bfsm_counters {
    double alg_bytes_per_eval;            /* (6*B_shard + 9) * G * c, the SURVEY 8(d) model                 */
    double kernel_ms[BFSM_K_COUNT];       /* summed HIP-event time of the last profiled evaluation         */
    double kernel_alg_bytes[BFSM_K_COUNT];/* algorithmic bytes moved by those launches                     */
    int kernel_launches[BFSM_K_COUNT];
    int n_chunks;
    int chunk_dirs;                       /* directions in the largest chunk                               */
    long long n_dirs;                     /* work units of this shard (directions; antipodal pairs if merged) */
    double moved_bytes_per_eval;          /* bytes the launch sequence moves (model); == alg bytes + slabs unless
                                             BFSM_FLAG_EXACT_REDUCTIONS is set                                */
    int exact_reductions;                 /* 1 if the flag is active                                          */
    int antipodal_merged;                 /* 1 if antipodal pairs were merged                                 */
} bfsm_counters;

/* == BoltzmannOperator<CUDA_Backend>::initialize() (CUDABoltzmannOperator.cu:28-115): allocates all device
 * scratch, uploads quadrature-derived tables, builds twiddles.  Returns a handle through *out. */
int bfsm_create(const bfsm_desc* desc, bfsm_handle* out);

/* == BoltzmannOperator<CUDA_Backend>::computeCollision(Q, f_in) (CUDABoltzmannOperator.cu:119-220), blocking:
 * returns after the device work has completed (the reference ends in cudaDeviceSynchronize, cu:218).
 * Requires a handle that owns ALL directions (dir_begin,dir_end = 0,0 or 0,n_gl*n_sph). */
int bfsm_collide(bfsm_handle h, double* Q_dev, const double* f_dev);

/* Same, enqueued on `stream` (a hipStream_t cast to void*; NULL = the device's legacy default stream, which is what
 * the reference launches on, cu:131-218) without the final host synchronisation. */
int bfsm_collide_async(bfsm_handle h, double* Q_dev, const double* f_dev, void* stream);

/* Batch of distributions (new functionality): f_dev and Q_dev hold n_batch <= desc.max_batch
 * consecutive N^3 arrays; every kernel launch covers the whole batch (one more grid dimension), so the quadrature
 * tables are shared and small grids (N = 16, 32) fill the GPU.  Member i of the result is bitwise identical to
 * bfsm_collide on member i alone ON THE SAME HANDLE (a handle created with max_batch > 1 keeps one set of kernels for
 * single and batched calls; a batch of one on a max_batch <= 1 handle is bfsm_collide itself).  Batches are
 * independent: on several GPUs they shard without any collective. */
int bfsm_collide_batch(bfsm_handle h, double* Q_dev, const double* f_dev, int n_batch);
int bfsm_collide_batch_async(bfsm_handle h, double* Q_dev, const double* f_dev, int n_batch, void* stream);
/* Batch x direction shard in one call (the two data-parallel axes composed): the handle may own any shard of the
 * directions; member i of Q_dev receives Re IFFT(this shard's partial Q_gain_hat of member i) [- loss term of member i
 * if with_loss].  The caller sums Q_dev (n_batch * G doubles) over the ranks with ONE collective for the whole batch;
 * exactly one rank passes with_loss != 0.  bfsm_collide_batch_async == this with all directions and with_loss = 1. */
int bfsm_collide_batch_partial_async(bfsm_handle h, double* Q_dev, const double* f_dev, int n_batch, int with_loss,
                                     void* stream);

/*
 * Sharded evaluation (new functionality: the reference is single-device).  Every rank calls
 *   bfsm_gain_partial()  -> this shard's partial Q_gain_hat in the handle-owned buffer bfsm_qhat_buffer()
 *   <one sum all-reduce / reduce of that buffer: RCCL over xGMI, by the caller>
 *   bfsm_finish()        -> loss term, final inverse transforms, Q        (cu:193-216)
 * bfsm_collide() == gain_partial + finish on one device.
 */
int bfsm_gain_partial(bfsm_handle h, const double* f_dev, void* stream);
int bfsm_finish(bfsm_handle h, double* Q_dev, const double* f_dev, void* stream);
/* Cheaper sharded route (half the bytes on the wire, nothing after the collective): the inverse transform is linear,
 * so every rank transforms its OWN partial Q_gain_hat and the caller sums the real results:
 *   bfsm_gain_partial()
 *   bfsm_finish_partial(h, Q, f, with_loss = (rank == 0), stream)   -> Q = Re IFFT(partial Q_gain_hat) [- loss term]
 *   <one sum all-reduce of Q (G doubles), by the caller>
 * Exactly one rank passes with_loss != 0. */
int bfsm_finish_partial(bfsm_handle h, double* Q_dev, const double* f_dev, int with_loss, void* stream);
/* The same two steps as ONE call, which lets the library fuse the slab reduce into the first tail kernel when the
 * shard has few slabs (one launch and one pass over Q_hat fewer; bitwise the same Q).  The contents of
 * bfsm_qhat_buffer() are unspecified after this call and after bfsm_collide / bfsm_collide_batch, which are this
 * sequence with with_loss = 1; only bfsm_gain_partial defines them. */
int bfsm_collide_partial_async(bfsm_handle h, double* Q_dev, const double* f_dev, int with_loss, void* stream);
/* Device pointer to the (partial) Q_gain_hat: n_elems reals of `precision` bits (2*G, interleaved complex in the
 * library's spectral layout [lx][lz][ly]); the buffer the collective must sum in place. */
void* bfsm_qhat_buffer(bfsm_handle h, size_t* n_elems, int* precision);

/* Blocks until everything enqueued by this handle has completed: the work of every stream that was passed to one of
 * its entry points since the previous bfsm_synchronize is waited for (through events the handle recorded itself), not
 * only the most recent one; the 64 most recently added distinct streams are tracked.  The list is cleared whether or
 * not the wait succeeds, so a failure does not affect later calls. */
int bfsm_synchronize(bfsm_handle h);

/* Batched 3-D complex transform with the library's own kernels (counterpart of the cufftPlanMany plan,
 * CUDABoltzmannOperator.cu:88-100; used by the FFT unit tests that mirror cufft_benchmark.cu:150-207).
 * data_dev: batch * G interleaved complex of the handle's precision, transformed in place, unnormalised.
 * sign -1 = forward: physical [x][y][z] in, spectral-transposed [lx][lz][ly] out (the fused pipeline's spectral layout);
 * sign +1 = backward: [lx][lz][ly] in, [x][y][z] out.  1 <= batch <= 65535 (one grid dimension).
 * On boxes served by the size-generic path the spectral side is the natural [lx][ly][lz]. */
int bfsm_fft3d(bfsm_handle h, void* data_dev, int batch, int sign);

int bfsm_get_counters(bfsm_handle h, bfsm_counters* out);

/* == ~BoltzmannOperator() (CUDABoltzmannOperator.cu:224-261) */
int bfsm_destroy(bfsm_handle h);

/* Message of the last failure on this handle (or of the last failed bfsm_create when h == NULL). */
const char* bfsm_last_error(bfsm_handle h);

/* "HIP" -- what getBackendName() returns (CUDABoltzmannOperator.hpp:60-62 returns "CUDA"). */
const char* bfsm_backend_name(void);
int bfsm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BFSM_H */
