// maxwell_bkw_hip_multi -- the BKW driver on P MI355X GPUs of one node, single process, RCCL over xGMI.
//
// New functionality (the reference is single-device): the B = M_gl * M_sph quadrature directions are sharded
// contiguously over the devices; per evaluation every device computes its partial Q_gain_hat
// (bfsm_gain_partial), inverse-transforms it (bfsm_finish_partial; both in one bfsm_collide_partial_async call, device 0
// also subtracts the loss term) and ONE
// grouped ncclAllReduce sums the real Q (G doubles) on all devices.  Same flags and report as maxwell_bkw_hip
// (reference: maxwell_bkw_cuda.cu:27-51,137-180) plus --gpus P.  bench.py does the same with one process per GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "Collisions/HIPBoltzmannOperator.hpp"
#include "Quadratures/GaussLegendre.hpp"
#include "Quadratures/SphericalDesign.hpp"
#include "Utilities/constants.hpp"
#include "Utilities/statistics.hpp"

#define HIP_OR_DIE(call)                                                                                          \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            std::cerr << "HIP Error: " << hipGetErrorString(e_) << " at " << __FILE__ << ":" << __LINE__ << "\n"; \
            std::exit(EXIT_FAILURE);                                                                              \
        }                                                                                                         \
    } while (0)
#define RCCL_OR_DIE(call)                                                                                          \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) {                                                                                  \
            std::cerr << "RCCL Error: " << ncclGetErrorString(r_) << " at " << __FILE__ << ":" << __LINE__ << "\n"; \
            std::exit(EXIT_FAILURE);                                                                              \
        }                                                                                                         \
    } while (0)

int main(int argc, char** argv) {
    int Nv = 64, Ns = 48, Ngl = 16, trials = 5, gpus = 1, warmup = 2;
    bool exact = false, hermitian = false;
    std::string design_dir;
    for (int i = 1; i < argc; ++i) {
        auto val = [&](const char* name) -> const char* {
            if (std::strcmp(argv[i], name) != 0) return nullptr;
            if (i + 1 >= argc) { std::cerr << "error: missing value for " << name << "\n"; std::exit(EXIT_FAILURE); }
            return argv[++i];
        };
        const char* v;
        if ((v = val("--Nv"))) Nv = std::atoi(v);
        else if ((v = val("--Ns"))) Ns = std::atoi(v);
        else if ((v = val("--Ngl"))) Ngl = std::atoi(v);
        else if ((v = val("-t")) || (v = val("--trials"))) trials = std::atoi(v);
        else if ((v = val("--gpus"))) gpus = std::atoi(v);
        else if ((v = val("--warmup"))) warmup = std::atoi(v);
        else if ((v = val("--design-dir"))) design_dir = v;
        else if (std::strcmp(argv[i], "--exact-reductions") == 0) exact = true;
        else if (std::strcmp(argv[i], "--hermitian") == 0) exact = hermitian = true;
        else { std::cerr << "error: unknown argument " << argv[i] << "\n"; return EXIT_FAILURE; }
    }
    int ndev = 0;
    HIP_OR_DIE(hipGetDeviceCount(&ndev));
    if (gpus < 1 || gpus > ndev) {
        std::cerr << "error: --gpus " << gpus << " but " << ndev << " device(s) visible\n";
        return EXIT_FAILURE;
    }
    std::cout << "\nRun arguments:\nNv = " << Nv << "\nNs = " << Ns << "\nNgl = " << Ngl << "\ntrials = " << trials
              << "\ngpus = " << gpus << "\n";
    if (!design_dir.empty()) SphericalDesign::setDataDirectory(design_dir);

    // BKW problem (maxwell_bkw_cuda.cu:58-107)
    const double gamma = 0, b_gamma = 1 / (4 * pi), S = 5, R = 2 * S, L = ((3 + std::sqrt(2.0)) / 2) * S;
    const double dv = 2 * L / Nv, t = 6.5, K = 1 - std::exp(-t / 6), dK = std::exp(-t / 6) / 6;
    const size_t G = static_cast<size_t>(Nv) * Nv * Nv;
    std::vector<double> f_h(G), Q_exact(G), Q_h(G);
    const double norm = 1 / (2 * std::pow(2 * pi * K, 1.5));
    for (int i = 0; i < Nv; ++i)
        for (int j = 0; j < Nv; ++j)
            for (int k = 0; k < Nv; ++k) {
                const double vx = -L + dv / 2 + i * dv, vy = -L + dv / 2 + j * dv, vz = -L + dv / 2 + k * dv;
                const double r2 = vx * vx + vy * vy + vz * vz, gauss = std::exp(-r2 / (2 * K));
                const double fv = norm * gauss * ((5 * K - 3) / K + (1 - K) / (K * K) * r2);
                const size_t idx = (static_cast<size_t>(i) * Nv + j) * Nv + k;
                f_h[idx] = fv;
                Q_exact[idx] = dK * ((-3 / (2 * K) + r2 / (2 * K * K)) * fv + norm * gauss * (3 / (K * K) + (K - 2) / (K * K * K) * r2));
            }

    auto gl = std::make_shared<GaussLegendreQuadrature>(Ngl, 0, R);
    auto sph = std::make_shared<SphericalDesign>(Ns);
    const long long B = static_cast<long long>(Ngl) * Ns;

    std::vector<int> devs(gpus);
    for (int g = 0; g < gpus; ++g) devs[g] = g;
    std::vector<ncclComm_t> comms(gpus);
    RCCL_OR_DIE(ncclCommInitAll(comms.data(), gpus, devs.data()));

    std::vector<double*> f_d(gpus), Q_d(gpus);
    std::vector<hipStream_t> streams(gpus);
    std::vector<std::unique_ptr<BoltzmannOperator<HIP_Backend>>> ops(gpus);
    using clk = std::chrono::steady_clock;
    const auto t_init = clk::now();
    for (int g = 0; g < gpus; ++g) {
        HIP_OR_DIE(hipSetDevice(g));
        HIP_OR_DIE(hipStreamCreate(&streams[g]));
        HIP_OR_DIE(hipMalloc(reinterpret_cast<void**>(&f_d[g]), G * sizeof(double)));
        HIP_OR_DIE(hipMalloc(reinterpret_cast<void**>(&Q_d[g]), G * sizeof(double)));
        HIP_OR_DIE(hipMemcpy(f_d[g], f_h.data(), G * sizeof(double), hipMemcpyHostToDevice));
        ops[g] = std::make_unique<BoltzmannOperator<HIP_Backend>>(gl, sph, Nv, Nv, Nv, gamma, b_gamma, L);
        ops[g]->setDevice(g);
        ops[g]->setExactReductions(exact, hermitian);
        const long long base = B / gpus, rem = B % gpus;        // contiguous, balanced shards
        const long long b0 = g * base + std::min<long long>(g, rem), b1 = b0 + base + (g < rem ? 1 : 0);
        ops[g]->setDirectionShard(b0, b1);
        ops[g]->initialize();
    }
    std::cout << "Initialization time (s): " << std::chrono::duration<double>(clk::now() - t_init).count() << " seconds\n";

    auto evaluate = [&]() {
        for (int g = 0; g < gpus; ++g) {
            ops[g]->collidePartial(Q_d[g], f_d[g], g == 0, streams[g]);   // gain_partial + finish_partial, fused
        }
        RCCL_OR_DIE(ncclGroupStart());
        for (int g = 0; g < gpus; ++g)
            RCCL_OR_DIE(ncclAllReduce(Q_d[g], Q_d[g], G, ncclDouble, ncclSum, comms[g], streams[g]));   // the ONE collective
        RCCL_OR_DIE(ncclGroupEnd());
        for (int g = 0; g < gpus; ++g) {
            HIP_OR_DIE(hipSetDevice(g));
            HIP_OR_DIE(hipStreamSynchronize(streams[g]));
        }
    };
    for (int w = 0; w < warmup; ++w) evaluate();
    std::vector<double> times;
    for (int trial = 0; trial < trials; ++trial) {
        const auto t0 = clk::now();
        evaluate();
        times.push_back(std::chrono::duration<double>(clk::now() - t0).count());
    }
    print_stats_summary("HIP x" + std::to_string(gpus), times);

    HIP_OR_DIE(hipSetDevice(gpus - 1));       // every device holds the full answer; check the last one
    HIP_OR_DIE(hipMemcpy(Q_h.data(), Q_d[gpus - 1], G * sizeof(double), hipMemcpyDeviceToHost));
    double err_L1 = 0, err_L2 = 0, err_Linf = 0;
    for (size_t i = 0; i < G; ++i) {
        const double d = std::abs(Q_h[i] - Q_exact[i]);
        err_L1 += d;
        err_L2 += d * d;
        err_Linf = std::max(err_Linf, d);
    }
    std::cout << "Approximation errors:\nL1 error: " << err_L1 * dv * dv * dv << "\nL2 error: " << std::sqrt(err_L2 * dv * dv * dv)
              << "\nLinf error: " << err_Linf << "\n\n";
    const RunStats st = summarize(times);
    std::cout << "{\"backend\": \"HIP\", \"n_gpus\": " << gpus << ", \"Nv\": " << Nv << ", \"Ngl\": " << Ngl << ", \"Ns\": " << Ns
              << ", \"evals_per_s\": " << 1.0 / st.mean << ", \"alg_GBps\": " << (6.0 * B + 9) * G * 16.0 / st.mean / 1e9 << "}\n";

    for (int g = 0; g < gpus; ++g) {
        HIP_OR_DIE(hipSetDevice(g));
        ops[g].reset();
        HIP_OR_DIE(hipFree(f_d[g]));
        HIP_OR_DIE(hipFree(Q_d[g]));
        HIP_OR_DIE(hipStreamDestroy(streams[g]));
        ncclCommDestroy(comms[g]);
    }
    return 0;
}
