// Streaming ceiling of KB's access pattern (two arrays read, one written in place; x-lines of N points at stride N*N, NPL
// contiguous columns per workgroup), no LDS, no butterflies: what does the memory system give this pattern?
//   case A: N = 128, fp32 complex (8 B), NPL = 64 columns  -> 512-byte runs   (config 5's KB)
//   case B: N = 128, fp32 complex, NPL = 128 columns        -> 1-KiB runs
//   case C: N = 64,  fp64 complex (16 B), NPL = 64 columns  -> 1-KiB runs     (config 3's KB)
// build: hipcc -O3 --offload-arch=gfx950 -o rw_stream rw_stream.hip ; run: ./rw_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

template <typename V, int N, int NPL, int E>
__global__ void __launch_bounds__(NPL * (N / E)) probe(V* a1, const V* a2) {
    constexpr int TT = N / E;
    const int tid = threadIdx.x, p = tid % NPL, u = tid / NPL;
    const size_t base = (size_t)blockIdx.y * N * N * N + (size_t)blockIdx.x * NPL + p;
    V a[E], b[E];
#pragma unroll
    for (int m = 0; m < E; ++m) a[m] = __builtin_nontemporal_load(a1 + base + (size_t)(u + TT * m) * N * N);
#pragma unroll
    for (int m = 0; m < E; ++m) b[m] = __builtin_nontemporal_load(a2 + base + (size_t)(u + TT * m) * N * N);
#pragma unroll
    for (int m = 0; m < E; ++m) __builtin_nontemporal_store(a[m] * b[m], a1 + base + (size_t)(u + TT * m) * N * N);
}

template <typename V, int N, int NPL, int E>
void run(const char* name, int n_dir) {
    const size_t elems = (size_t)n_dir * N * N * N;
    V *a1, *a2;
    CHECK(hipMalloc(&a1, elems * sizeof(V)));
    CHECK(hipMalloc(&a2, elems * sizeof(V)));
    CHECK(hipMemset(a1, 0, elems * sizeof(V)));
    CHECK(hipMemset(a2, 0, elems * sizeof(V)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid(N * N / NPL, n_dir);
    probe<V, N, NPL, E><<<grid, NPL * (N / E)>>>(a1, a2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) probe<V, N, NPL, E><<<grid, NPL * (N / E)>>>(a1, a2);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double gb = 3.0 * elems * sizeof(V) / 1e9;
    std::printf("%-62s %8.3f ms  %6.2f TB/s  (%.2f GB)\n", name, ms, gb / ms, gb);
    CHECK(hipFree(a1)); CHECK(hipFree(a2));
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d2 __attribute__((ext_vector_type(2)));

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<f2, 128, 64, 16>("A  N=128 fp32, 64 columns per workgroup (512-B runs), 512 thr", 768);
        run<f2, 128, 128, 16>("B  N=128 fp32, 128 columns per workgroup (1-KiB runs), 1024 thr", 768);
        run<d2, 64, 64, 8>("C  N=64 fp64, 64 columns per workgroup (1-KiB runs), 512 thr", 768);
    }
    return 0;
}
