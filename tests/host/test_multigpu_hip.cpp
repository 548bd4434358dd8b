// TEST ONLY (GPU).  BoltzmannOperator<HIP_MultiGPU_Backend> on real HIP + RCCL with the devices that are visible
// (one on the development box: the collectives are forced on, communicator of size 1):
//   * computeCollisionBatch(n) equals n single evaluations bit for bit;
//   * f produced on a NON-BLOCKING stream and named with setInputStream(): the broadcast waits for it (without the
//     call the operator's stream would be ordered behind the legacy default stream only);
//   * counters() of every device of the team; setMaxChunk() reaches the per-device operators.
// Built and run by tests/test_gpu_parity.py::test_cpp_multi_gpu_operator_batches_and_input_stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "Collisions/HIPMultiGPUBoltzmannOperator.hpp"
#include "Utilities/constants.hpp"

#define OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); ++failures; } } while (0)

__global__ void spin(long long cycles, double* sink) {          // keeps a stream busy for a while
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 1000) *sink = 1;
}

int main(int argc, char** argv) {
    if (argc > 1) SphericalDesign::setDataDirectory(argv[1]);
    int ndev = 0;
    OK(hipGetDeviceCount(&ndev));
    const int P = ndev >= 2 ? 2 : 1;
    const int Nv = 32, Ngl = 4, Ns = 12, NB = 3;
    const size_t G = (size_t)Nv * Nv * Nv;
    const double gamma = 0, b_gamma = 1 / (4 * pi), S = 5, R = 2 * S, L = ((3 + std::sqrt(2.0)) / 2) * S;
    std::vector<double> f_h(NB * G);
    for (int m = 0; m < NB; ++m)
        for (size_t i = 0; i < G; ++i) {
            const int x = (int)(i / (Nv * Nv)), y = (int)(i / Nv % Nv), z = (int)(i % Nv);
            const double r2 = std::pow((x - 15.5 - m) / 6.0, 2) + std::pow((y - 15.5) / 5.0, 2) + std::pow((z - 14.5 + m) / 7.0, 2);
            f_h[m * G + i] = std::exp(-r2) * (1.0 + 0.1 * m);
        }
    OK(hipSetDevice(0));
    double *f_d, *Q_d, *Q1_d;
    OK(hipMalloc((void**)&f_d, NB * G * sizeof(double)));
    OK(hipMalloc((void**)&Q_d, NB * G * sizeof(double)));
    OK(hipMalloc((void**)&Q1_d, NB * G * sizeof(double)));
    OK(hipMemcpy(f_d, f_h.data(), NB * G * sizeof(double), hipMemcpyHostToDevice));

    auto gl = std::make_shared<GaussLegendreQuadrature>(Ngl, 0, R);
    auto sph = std::make_shared<SphericalDesign>(Ns);
    BoltzmannOperator<HIP_MultiGPU_Backend> op(gl, sph, Nv, Nv, Nv, gamma, b_gamma, L);
    std::vector<int> devs;
    for (int g = 0; g < P; ++g) devs.push_back(g);
    op.setDevices(devs);
    op.setForceCollectives(true);
    op.setMaxBatch(NB);
    op.setMaxChunk(10);
    op.setProfiling(true);
    op.initialize();

    // batch == singles, bit for bit
    op.computeCollisionBatch(Q_d, f_d, NB);
    for (int m = 0; m < NB; ++m) op(Q1_d + m * G, f_d + m * G);
    std::vector<double> Qb(NB * G), Q1(NB * G);
    OK(hipMemcpy(Qb.data(), Q_d, NB * G * sizeof(double), hipMemcpyDeviceToHost));
    OK(hipMemcpy(Q1.data(), Q1_d, NB * G * sizeof(double), hipMemcpyDeviceToHost));
    CHECK(std::memcmp(Qb.data(), Q1.data(), NB * G * sizeof(double)) == 0);
    double amax = 0;
    for (double v : Q1) amax = std::max(amax, std::abs(v));
    CHECK(amax > 1e-6);

    // counters of every device: the shards tile the B directions, chunks of at most 10
    long long covered = 0;
    for (int g = 0; g < P; ++g) {
        const bfsm_counters c = op.counters(g);
        covered += c.n_dirs;
        CHECK(c.chunk_dirs <= 10 && c.n_chunks >= 1 && c.kernel_ms[BFSM_K_GAIN_LINE] > 0);
    }
    CHECK(covered == (long long)Ngl * Ns);

    // f produced on a non-blocking stream behind a long-running kernel
    hipStream_t prod;
    OK(hipStreamCreateWithFlags(&prod, hipStreamNonBlocking));
    double* stage;
    OK(hipMalloc((void**)&stage, G * sizeof(double)));
    OK(hipMemcpy(stage, f_h.data() + G, G * sizeof(double), hipMemcpyHostToDevice));      // member 1 of the batch
    OK(hipMemset(f_d, 0, G * sizeof(double)));
    OK(hipDeviceSynchronize());
    op.setInputStream(prod);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, prod, 200000000LL, (double*)nullptr);   // ~0.1 s
    OK(hipMemcpyAsync(f_d, stage, G * sizeof(double), hipMemcpyDeviceToDevice, prod));
    op(Q_d, f_d);                                        // must see member 1's f, not the zeros
    std::vector<double> Qs(G);
    OK(hipMemcpy(Qs.data(), Q_d, G * sizeof(double), hipMemcpyDeviceToHost));
    CHECK(std::memcmp(Qs.data(), Q1.data() + G, G * sizeof(double)) == 0);
    op.clearInputStream();
    OK(hipStreamSynchronize(prod));
    op(Q_d, f_d);
    OK(hipMemcpy(Qs.data(), Q_d, G * sizeof(double), hipMemcpyDeviceToHost));
    CHECK(std::memcmp(Qs.data(), Q1.data() + G, G * sizeof(double)) == 0);
    OK(hipStreamDestroy(prod));
    OK(hipFree(stage)); OK(hipFree(f_d)); OK(hipFree(Q_d)); OK(hipFree(Q1_d));
    if (failures == 0) std::printf("multi-GPU operator checks passed on %d device(s)\n", P);
    return failures == 0 ? 0 : 1;
}
