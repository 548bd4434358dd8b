// Streaming-store ceilings in KA's access pattern (N = 64 fp64): 512 workgroups of 512 threads, each writes 64 KiB tiles
// (64 rows of 1 KiB, 16 B per lane, thread (u, p) -> rows u + 8 m) into two arrays at a 4 MiB stride per direction, no
// arithmetic, no LDS.  What does the store stream alone reach, and does the cache-policy hint or the row order matter?
//   mode 0: plain global_store_dwordx4           mode 1: nontemporal (nt)        mode 2: sc0 sc1 nt      mode 3: sc1
//   mode 4: nt, rows in contiguous order per wave (wave w writes rows 8 w .. 8 w + 7: 8 KiB runs per wave)
//   mode 5: nt, a workgroup barrier after every tile (what the exchanges of the real kernel impose)
// build: hipcc -O3 --offload-arch=gfx950 -o store_stream store_stream.hip ; run: ./store_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef d2 __attribute__((address_space(1))) * gd2;

template <int MODE>
__device__ __forceinline__ void st(d2* p, d2 v) {
    if constexpr (MODE == 0) *p = v;
    else if constexpr (MODE == 1 || MODE >= 4) __builtin_nontemporal_store(v, p);
    else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ void __launch_bounds__(512, 4) probe(d2* a1, d2* a2, int per_group, int n_dir) {
    constexpr int N = 64;
    const int tid = threadIdx.x, p = tid % N, u = tid / N;
    const int lx = blockIdx.x, g = blockIdx.y;
    d2 v = {(double)tid, (double)lx};
    for (int i = 0; i < per_group; ++i) {
        const int d = g * per_group + i;
        if (d >= n_dir) break;
        for (int sgn = 0; sgn < 2; ++sgn) {
            d2* dst = (sgn ? a2 : a1) + ((size_t)d * N + lx) * N * N;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int row = (MODE == 4) ? (8 * u + m) : (u + 8 * m);
                st<MODE>(dst + (size_t)row * N + p, v);
            }
            v.x += 1.0;
            if (MODE == 5) __syncthreads();
        }
    }
}

template <int MODE>
double run(d2* a1, d2* a2, int n_dir) {
    const int groups = 8, per_group = (n_dir + groups - 1) / groups;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<MODE><<<dim3(64, groups), 512>>>(a1, a2, per_group, n_dir);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) probe<MODE><<<dim3(64, groups), 512>>>(a1, a2, per_group, n_dir);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    const int n_dir = 768;
    const size_t elems = (size_t)n_dir * 64 * 64 * 64;      // d2 elements per array (4 MiB per direction)
    d2 *a1, *a2;
    CHECK(hipMalloc(&a1, elems * sizeof(d2)));
    CHECK(hipMalloc(&a2, elems * sizeof(d2)));
    const double gb = 2.0 * elems * sizeof(d2) / 1e9;
    const char* names[] = {"plain", "nt", "sc0 sc1 nt", "sc1", "nt, contiguous rows per wave", "nt + barrier per tile"};
    double ms[6];
    for (int rep = 0; rep < 2; ++rep) {
        ms[0] = run<0>(a1, a2, n_dir); ms[1] = run<1>(a1, a2, n_dir); ms[2] = run<2>(a1, a2, n_dir);
        ms[3] = run<3>(a1, a2, n_dir); ms[4] = run<4>(a1, a2, n_dir); ms[5] = run<5>(a1, a2, n_dir);
        for (int m = 0; m < 6; ++m)
            std::printf("mode %d  %-30s %7.3f ms  %6.2f TB/s  (%.3f GB, KA's store stream at cfg3)\n", m, names[m], ms[m], gb / ms[m], gb);
    }
    return 0;
}
