// Read-only streaming ceilings in two access patterns of the pipeline (N = 64 fp64, 768 directions of 4 MiB, no LDS, no arithmetic
// beyond one add per element):
//   mode 0  KC's pattern: workgroup (plane x, group g) walks the directions of its group and reads the contiguous 64 KiB tile
//           [y][z] of plane x of each (512 threads, 8 x 16 B per thread and direction, rows u + 8 m)
//   mode 1  the same, but the workgroups of different planes start at different directions of the group (rotated start)
//   mode 2  KB's read pattern: workgroup (y, direction) reads the 64 x-rows of 1 KiB (stride 64 KiB) of ONE array
//   mode 3  KB's read pattern on TWO arrays (what KB reads)
//   mode 4  KC's pattern with the next direction's loads issued before the current ones are consumed (two directions in flight)
// build: hipcc -O3 --offload-arch=gfx950 -o read_stream read_stream.hip ; run: ./read_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int N = 64;

template <int MODE>
__global__ void __launch_bounds__(512, 4) probe(const d2* a1, const d2* a2, d2* out, int per_group, int n_dir) {
    const int tid = threadIdx.x, p = tid % N, u = tid / N;
    d2 acc = {0.0, 0.0};
    if constexpr (MODE == 0 || MODE == 1 || MODE == 4) {
        const int x = blockIdx.x, g = blockIdx.y;
        const int d0 = g * per_group;
        int n = per_group;
        if (d0 + n > n_dir) n = n_dir - d0;
        const int rot = (MODE == 1) ? (x * 3) % (n > 0 ? n : 1) : 0;
        if constexpr (MODE != 4) {
            for (int i = 0; i < n; ++i) {
                int d = d0 + i + rot;
                if (d >= d0 + n) d -= n;
                const d2* src = a1 + ((size_t)d * N + x) * N * N;
#pragma unroll
                for (int m = 0; m < 8; ++m) acc += __builtin_nontemporal_load(src + (size_t)(u + 8 * m) * N + p);
            }
        } else {
            d2 v[8], w[8];
            const d2* src = a1 + ((size_t)d0 * N + x) * N * N;
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = __builtin_nontemporal_load(src + (size_t)(u + 8 * m) * N + p);
            for (int i = 0; i < n; ++i) {
                const int dn = (i + 1 < n) ? d0 + i + 1 : d0 + i;
                const d2* s2 = a1 + ((size_t)dn * N + x) * N * N;
#pragma unroll
                for (int m = 0; m < 8; ++m) w[m] = __builtin_nontemporal_load(s2 + (size_t)(u + 8 * m) * N + p);
#pragma unroll
                for (int m = 0; m < 8; ++m) acc += v[m];
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = w[m];
            }
        }
        if (acc.x == 12345.678) out[blockIdx.x] = acc;
    } else {
        const int y = blockIdx.x, d = blockIdx.y;
        const size_t base = (size_t)d * N * N * N + (size_t)y * N + p;
#pragma unroll
        for (int m = 0; m < 8; ++m) acc += __builtin_nontemporal_load(a1 + base + (size_t)(u + 8 * m) * N * N);
        if constexpr (MODE == 3) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc += __builtin_nontemporal_load(a2 + base + (size_t)(u + 8 * m) * N * N);
        }
        if (acc.x == 12345.678) out[blockIdx.x] = acc;
    }
}

template <int MODE>
double run(const d2* a1, const d2* a2, d2* out, int n_dir, int groups = 8) {
    const int per_group = (n_dir + groups - 1) / groups;
    const dim3 grid = (MODE == 0 || MODE == 1 || MODE == 4) ? dim3(N, groups) : dim3(N, n_dir);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<MODE><<<grid, 512>>>(a1, a2, out, per_group, n_dir);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) probe<MODE><<<grid, 512>>>(a1, a2, out, per_group, n_dir);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    const int n_dir = 768;
    const size_t elems = (size_t)n_dir * N * N * N;
    d2 *a1, *a2, *out;
    CHECK(hipMalloc(&a1, elems * sizeof(d2)));
    CHECK(hipMalloc(&a2, elems * sizeof(d2)));
    CHECK(hipMalloc(&out, 4096 * sizeof(d2)));
    CHECK(hipMemset(a1, 0, elems * sizeof(d2)));
    CHECK(hipMemset(a2, 0, elems * sizeof(d2)));
    const double gb1 = elems * sizeof(d2) / 1e9;
    const char* names[] = {"KC pattern (64 KiB tiles, 4 MiB stride)", "KC pattern, rotated start per plane", "KB read pattern, one array",
                           "KB read pattern, two arrays", "KC pattern, two directions in flight"};
    for (int rep = 0; rep < 2; ++rep) {
        double ms[5] = {run<0>(a1, a2, out, n_dir), run<1>(a1, a2, out, n_dir), run<2>(a1, a2, out, n_dir), run<3>(a1, a2, out, n_dir),
                        run<4>(a1, a2, out, n_dir)};
        for (int m = 0; m < 5; ++m) {
            const double gb = (m == 3) ? 2 * gb1 : gb1;
            std::printf("mode %d  %-42s %7.3f ms  %6.2f TB/s  (%.2f GB)\n", m, names[m], ms[m], gb / ms[m], gb);
        }
        // KC's own grid at config 3: 16 groups of 48 directions = 1024 workgroups = two rounds of 512
        const double t16 = run<0>(a1, a2, out, n_dir, 16), t32 = run<0>(a1, a2, out, n_dir, 32);
        std::printf("mode 0  %-42s %7.3f ms  %6.2f TB/s\n", "KC pattern, 16 groups (two rounds)", t16, gb1 / t16);
        std::printf("mode 0  %-42s %7.3f ms  %6.2f TB/s\n", "KC pattern, 32 groups (four rounds)", t32, gb1 / t32);
    }
    return 0;
}
