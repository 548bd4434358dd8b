// maxwell_bkw_hip_multi -- the BKW driver on P MI355X GPUs of one node, single process, RCCL over xGMI.
//
// The reference's driver (maxwell_bkw_cuda.cu:27-51,137-180) with the backend tag changed to HIP_MultiGPU_Backend:
// f and Q live on the first device, `collision_operator(Q, f)` is one blocking call, and the operator class
// (Collisions/HIPMultiGPUBoltzmannOperator.hpp) shards the B = M_gl * M_sph quadrature directions over the devices,
// broadcasts f, and sums the partial results with ONE grouped RCCL reduce.  Same flags and report as maxwell_bkw_hip
// (--Nv --Ns --Ngl -t/--trials --warmup --precision {64,32} --input {bkw,random} --design-dir --exact-reductions
// --hermitian) plus --gpus P | --devices 0,1,...  (the first entry owns f and Q), --chunk N (directions resident at once
// per device), --force-rccl (use the collectives even with one device), --counters (per-device kernel times of the last
// evaluation).  BASELINE config 5 in its 8-GPU form:  --Nv 128 --Ngl 30 --Ns 192 --precision 32 --gpus 8.
// bench.py does the same with one process per GPU.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "Collisions/HIPMultiGPUBoltzmannOperator.hpp"
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
namespace {
// u in [0,1) from splitmix64(index + seed): the seeded perturbation of maxwell_bkw_hip --input random
double unit_random(std::uint64_t idx, std::uint64_t seed) {
    std::uint64_t z = idx + seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0);
}
}  // namespace

int main(int argc, char** argv) {
    int Nv = 64, Ns = 48, Ngl = 16, trials = 5, gpus = -1, warmup = 2, precision = 64, chunk = 0;
    bool exact = false, hermitian = false, force_rccl = false, show_counters = false;
    std::string design_dir, input = "bkw", device_list;
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
        else if ((v = val("--precision"))) precision = std::atoi(v);
        else if ((v = val("--chunk"))) chunk = std::atoi(v);
        else if ((v = val("--input"))) input = v;
        else if ((v = val("--devices"))) device_list = v;
        else if (std::strcmp(argv[i], "--counters") == 0) show_counters = true;
        else if (std::strcmp(argv[i], "--exact-reductions") == 0) exact = true;
        else if (std::strcmp(argv[i], "--hermitian") == 0) exact = hermitian = true;
        else if (std::strcmp(argv[i], "--force-rccl") == 0) force_rccl = true;
        else { std::cerr << "error: unknown argument " << argv[i] << "\n"; return EXIT_FAILURE; }
    }
    int ndev = 0;
    HIP_OR_DIE(hipGetDeviceCount(&ndev));
    std::vector<int> devs;
    if (!device_list.empty()) {                        // --devices 2,0,5: ordinals, the first one owns f and Q
        size_t pos = 0;
        while (pos <= device_list.size()) {
            const size_t comma = device_list.find(',', pos);
            const std::string tok = device_list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            char* end = nullptr;
            const long d = std::strtol(tok.c_str(), &end, 10);
            if (tok.empty() || *end != '\0') { std::cerr << "error: --devices expects a comma-separated list of ordinals\n"; return EXIT_FAILURE; }
            devs.push_back(static_cast<int>(d));
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
        if (gpus >= 0 && gpus != static_cast<int>(devs.size())) { std::cerr << "error: --gpus contradicts --devices\n"; return EXIT_FAILURE; }
        gpus = static_cast<int>(devs.size());
    } else {
        if (gpus < 0) gpus = 1;
        for (int g = 0; g < gpus; ++g) devs.push_back(g);
    }
    if (gpus < 1 || gpus > ndev) {
        std::cerr << "error: --gpus " << gpus << " but " << ndev << " device(s) visible\n";
        return EXIT_FAILURE;
    }
    for (size_t g = 0; g < devs.size(); ++g)
        for (size_t h = 0; h <= g; ++h)
            if (devs[g] < 0 || devs[g] >= ndev || (h < g && devs[h] == devs[g])) {
                std::cerr << "error: --devices must name distinct visible devices (" << ndev << " visible)\n";
                return EXIT_FAILURE;
            }
    if (precision != 64 && precision != 32) { std::cerr << "error: --precision must be 64 or 32\n"; return EXIT_FAILURE; }
    if (input != "bkw" && input != "random") { std::cerr << "error: --input must be bkw or random\n"; return EXIT_FAILURE; }
    std::cout << "\nRun arguments:\nNv = " << Nv << "\nNs = " << Ns << "\nNgl = " << Ngl << "\ntrials = " << trials
              << "\ngpus = " << gpus << "\n";
    if (precision != 64) std::cout << "precision = " << precision << "\n";
    if (!device_list.empty()) std::cout << "devices = " << device_list << "\n";
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

    const bool random_input = input == "random";
    if (random_input)
        for (size_t i = 0; i < G; ++i) f_h[i] *= 1.0 + 0.1 * unit_random(i, 0x5EED);
    HIP_OR_DIE(hipSetDevice(devs[0]));
    double *f_d = nullptr, *Q_d = nullptr;            // on the first device, like the reference's driver (cu:119-126)
    HIP_OR_DIE(hipMalloc(reinterpret_cast<void**>(&f_d), G * sizeof(double)));
    HIP_OR_DIE(hipMalloc(reinterpret_cast<void**>(&Q_d), G * sizeof(double)));
    HIP_OR_DIE(hipMemcpy(f_d, f_h.data(), G * sizeof(double), hipMemcpyHostToDevice));

    using clk = std::chrono::steady_clock;
    const auto t_init = clk::now();
    BoltzmannOperator<HIP_MultiGPU_Backend> collision_operator(gl, sph, Nv, Nv, Nv, gamma, b_gamma, L);
    collision_operator.setDevices(devs);
    collision_operator.setPrecision(precision);
    collision_operator.setMaxChunk(chunk);
    collision_operator.setProfiling(show_counters);
    collision_operator.setExactReductions(exact, hermitian);
    collision_operator.setForceCollectives(force_rccl);
    collision_operator.initialize();
    std::cout << "Initialization time (s): " << std::chrono::duration<double>(clk::now() - t_init).count() << " seconds\n";

    auto evaluate = [&]() { collision_operator(Q_d, f_d); };   // blocking
    for (int w = 0; w < warmup; ++w) evaluate();
    std::vector<double> times;
    for (int trial = 0; trial < trials; ++trial) {
        const auto t0 = clk::now();
        evaluate();
        times.push_back(std::chrono::duration<double>(clk::now() - t0).count());
    }
    print_stats_summary("HIP x" + std::to_string(gpus), times);

    HIP_OR_DIE(hipMemcpy(Q_h.data(), Q_d, G * sizeof(double), hipMemcpyDeviceToHost));
    if (!random_input) {
        double err_L1 = 0, err_L2 = 0, err_Linf = 0;
        for (size_t i = 0; i < G; ++i) {
            const double d = std::abs(Q_h[i] - Q_exact[i]);
            err_L1 += d;
            err_L2 += d * d;
            err_Linf = std::max(err_Linf, d);
        }
        std::cout << "Approximation errors:\nL1 error: " << err_L1 * dv * dv * dv << "\nL2 error: " << std::sqrt(err_L2 * dv * dv * dv)
                  << "\nLinf error: " << err_Linf << "\n\n";
    } else {
        double sabs = 0;
        for (size_t i = 0; i < G; ++i) sabs += std::abs(Q_h[i]);
        std::cout << "sum |Q| = " << std::scientific << std::setprecision(10) << sabs << "\n\n";
    }
    if (show_counters)                                 // per device: its shard and the kernel times of the last evaluation
        for (int g = 0; g < gpus; ++g) {
            const bfsm_counters c = collision_operator.counters(g);
            double ms = 0;
            for (int k = 0; k < BFSM_K_COUNT; ++k) ms += c.kernel_ms[k];
            std::cout << std::defaultfloat << std::setprecision(6) << "device " << collision_operator.devices()[g] << ": directions " << c.n_dirs << ", chunks " << c.n_chunks
                      << " x " << c.chunk_dirs << ", kernels " << ms << " ms (gain_inv " << c.kernel_ms[BFSM_K_GAIN_INV] << ", gain_line "
                      << c.kernel_ms[BFSM_K_GAIN_LINE] << ", gain_fwd " << c.kernel_ms[BFSM_K_GAIN_FWD] << ")\n";
        }
    const RunStats st = summarize(times);
    const double cbytes = precision == 64 ? 16.0 : 8.0;
    std::cout << std::defaultfloat << std::setprecision(6) << "{\"backend\": \"HIP\", \"n_gpus\": " << gpus << ", \"Nv\": " << Nv << ", \"Ngl\": " << Ngl << ", \"Ns\": " << Ns
              << ", \"precision\": " << precision << ", \"evals_per_s\": " << 1.0 / st.mean << ", \"alg_GBps\": " << (6.0 * B + 9) * G * cbytes / st.mean / 1e9 << "}\n";

    HIP_OR_DIE(hipFree(f_d));
    HIP_OR_DIE(hipFree(Q_d));
    return 0;
}
