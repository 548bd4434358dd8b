// Does fp64 VALU work overlap with LDS exchange traffic on a gfx950 CU?  (KA/KB/KC budget question)
// 512-thread workgroups, 64 KiB of LDS each (two per CU), 512 workgroups; per iteration a wave issues V fp64 FMAs and
// an 8-value b128 exchange (8 ds_write_b128 + 8 ds_read_b128) X times.
//   mode 0: VALU only          mode 1: LDS only (with the workgroup barriers of an exchange)
//   mode 2: both, in sequence in every wave, with barriers (what the FFT kernels do)
//   mode 3: both, no barriers
//   mode 4: specialised waves: even waves 2x VALU only, odd waves 2x LDS only (same totals per workgroup)
// build: hipcc -O3 --offload-arch=gfx950 -o valu_lds_overlap valu_lds_overlap.hip ; run: ./valu_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int V>
__device__ __forceinline__ void valu_block(double (&r)[16], double c) {
#pragma unroll
    for (int i = 0; i < V / 16; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = __builtin_fma(r[j], c, r[(j + 1) & 15]);
}

__device__ __forceinline__ void lds_exchange(d2* lds, d2 (&v)[8], int tid, bool barriers) {
    // write rows (k*8 + u) of a 64 x 65 tile, read the transposed pattern: conflict-free, like fft_line_np
    const int p = tid & 63, u = tid >> 6;
    if (barriers) __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) lds[(k * 8 + u) * 65 + p] = v[k];
    if (barriers) __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = lds[(u * 8 + k) * 65 + p];
}

template <int MODE>
__global__ void __launch_bounds__(512, 4) probe(double* out, int iters, double c) {
    extern __shared__ __align__(16) unsigned char smem[];
    d2* lds = reinterpret_cast<d2*>(smem);
    const int tid = threadIdx.x, wave = tid >> 6;
    double r[16];
    d2 v[8];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = tid * 1e-3 + j;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = d2{(double)tid, (double)k};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { valu_block<336>(r, c); }
        else if (MODE == 1) { for (int x = 0; x < 3; ++x) lds_exchange(lds, v, tid, true); }
        else if (MODE == 2) { for (int x = 0; x < 3; ++x) { valu_block<112>(r, c); lds_exchange(lds, v, tid, true); } }
        else if (MODE == 3) { for (int x = 0; x < 3; ++x) { valu_block<112>(r, c); lds_exchange(lds, v, tid, false); } }
        else {
            if (wave & 1) { for (int x = 0; x < 6; ++x) lds_exchange(lds, v, tid, false); }
            else { valu_block<672>(r, c); }
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += r[j];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
    if (s == 12345.678) out[0] = s;   // keep the work alive
}

template <int MODE>
float run(double* out, int iters) {
    const size_t lds = 64 * 65 * sizeof(d2);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<MODE>, dim3(512), dim3(512), lds, 0, out, iters, 1.0000001);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe<MODE>, dim3(512), dim3(512), lds, 0, out, iters, 1.0000001);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    double* out;
    CHECK(hipMalloc(&out, 8));
    const int iters = 192;   // KA at cfg3: 192 signed directions per workgroup
    const float t0 = run<0>(out, iters), t1 = run<1>(out, iters), t2 = run<2>(out, iters), t3 = run<3>(out, iters), t4 = run<4>(out, iters);
    std::printf("iters %d: VALU only %.3f ms | LDS only %.3f ms | both+barriers %.3f ms | both, no barriers %.3f ms | specialised waves %.3f ms\n",
                iters, t0, t1, t2, t3, t4);
    std::printf("sum VALU+LDS = %.3f ms, max = %.3f ms\n", t0 + t1, t0 > t1 ? t0 : t1);
    return 0;
}
