// VALU issue-rate microbenchmark: cycles per wave-instruction per SIMD for the operations the FFT butterflies use,
// at 1, 2 and 4 waves per SIMD (one workgroup per CU, every CU busy).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
    float a[16];
    v2f p[16];
    double d[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = {a[i], a[i] + 1.f}; d[i] = a[i]; }
    const float c = 1.0001f, e = 1e-7f;
    const v2f pc = {1.0001f, 0.9999f}, pe = {1e-7f, 2e-7f};
    const double dc = 1.0001, de = 1e-7;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
            if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pe));
            if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pe));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dc), "v"(de));
            if (OP == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(de));
            if (OP == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
            if (OP == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int OP>
void run(const char* name, float* out) {
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<OP><<<256, threads>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<256, threads>>>(out, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0, cyc = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
        const double n = 16.0 * iters;                       // instructions per wave
        const int wps = threads / 256;                       // waves per SIMD
        printf("%-14s waves/SIMD %d : %.2f cycles per wave-instruction per SIMD (memtime %.0f cycles, %.3f ms -> %.2f GHz)\n",
               name, wps, cyc / (n * wps), cyc, ms, cyc / (ms * 1e6));
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    run<0>("v_fma_f32", out); run<1>("v_add_f32", out); run<2>("v_mul_f32", out);
    run<3>("v_pk_fma_f32", out); run<4>("v_pk_add_f32", out); run<5>("v_pk_mul_f32", out);
    run<6>("v_fma_f64", out); run<7>("v_add_f64", out); run<8>("v_mul_f64", out); run<9>("v_mov_b32", out);
    return 0;
}
