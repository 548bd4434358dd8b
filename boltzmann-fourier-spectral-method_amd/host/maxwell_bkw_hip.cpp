// maxwell_bkw_hip -- BKW known-answer driver for the HIP backend on MI355X.
//
// Counterpart of the reference's maxwell_bkw_cuda.cu: same physics constants (cu:58-64), same BKW f / Q at t = 6.5
// (cu:81-107), same flags --Nv --Ns -t/--trials (cu:30-36), same printed report (run arguments, initialization
// time, timing statistics, L1/L2/Linf), device-resident f and Q around the timed loop (cu:119-156).
// Extra flags: --Ngl (the reference hard-wires M_gl = Nv, cu:110), --precision {64,32}, --input {bkw,random},
// --design-dir, --device, --warmup.  Linf is computed with a correct max-reduction (the reference's is not, cu:162-170).
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

#include "Collisions/HIPBoltzmannOperator.hpp"
#include "Quadratures/GaussLegendre.hpp"
#include "Quadratures/SphericalDesign.hpp"
#include "Utilities/constants.hpp"
#include "Utilities/statistics.hpp"

#define HIP_OR_DIE(call)                                                                             \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            std::cerr << "HIP Error: " << hipGetErrorString(e_) << " at " << __FILE__ << ":" << __LINE__ << std::endl; \
            std::exit(EXIT_FAILURE);                                                                 \
        }                                                                                            \
    } while (0)

namespace {
struct Args {
    int Nv = 32, Ns = 12, trials = 1, Ngl = -1, precision = 64, device = 0, warmup = 0;
    std::string input = "bkw", design_dir;
    bool exact = false, hermitian = false;   // opt-in exact work reductions (include/bfsm.h)
};

bool take(int& i, int argc, char** argv, const char* name, std::string& out) {
    if (std::strcmp(argv[i], name) != 0) return false;
    if (i + 1 >= argc) { std::cerr << "error: missing value for " << name << "\n"; std::exit(EXIT_FAILURE); }
    out = argv[++i];
    return true;
}

Args parse(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string v;
        if (take(i, argc, argv, "--Nv", v)) a.Nv = std::stoi(v);
        else if (take(i, argc, argv, "--Ns", v)) a.Ns = std::stoi(v);
        else if (take(i, argc, argv, "-t", v) || take(i, argc, argv, "--trials", v)) a.trials = std::stoi(v);
        else if (take(i, argc, argv, "--Ngl", v)) a.Ngl = std::stoi(v);
        else if (take(i, argc, argv, "--precision", v)) a.precision = std::stoi(v);
        else if (take(i, argc, argv, "--device", v)) a.device = std::stoi(v);
        else if (take(i, argc, argv, "--warmup", v)) a.warmup = std::stoi(v);
        else if (take(i, argc, argv, "--input", v)) a.input = v;
        else if (take(i, argc, argv, "--design-dir", v)) a.design_dir = v;
        else if (std::strcmp(argv[i], "--exact-reductions") == 0) a.exact = true;
        else if (std::strcmp(argv[i], "--hermitian") == 0) a.exact = a.hermitian = true;
        else { std::cerr << "error: unknown argument " << argv[i] << "\n"; std::exit(EXIT_FAILURE); }
    }
    if (a.Ngl < 0) a.Ngl = a.Nv;   // reference behaviour (maxwell_bkw_cuda.cu:110)
    return a;
}

// u in [0,1) from splitmix64(index + seed): the seeded perturbation used by the parity tests
double unit_random(std::uint64_t idx, std::uint64_t seed) {
    std::uint64_t z = idx + seed + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0);
}
}  // namespace

int main(int argc, char** argv) {
    const Args a = parse(argc, argv);
    const int Nv = a.Nv;
    std::cout << "\nRun arguments:\n"
              << "Nv = " << Nv << "\n"
              << "Ns = " << a.Ns << "\n"
              << "trials = " << a.trials << "\n";
    if (a.Ngl != Nv) std::cout << "Ngl = " << a.Ngl << "\n";
    if (!a.design_dir.empty()) SphericalDesign::setDataDirectory(a.design_dir);

    // Maxwell molecules and the spectral-method support constants
    const double gamma = 0;
    const double b_gamma = 1 / (4 * pi);
    const double S = 5, R = 2 * S, L = ((3 + std::sqrt(2.0)) / 2) * S;
    const double dv = 2 * L / Nv;
    std::vector<double> v(Nv);
    for (int i = 0; i < Nv; ++i) v[i] = -L + dv / 2 + i * dv;

    // BKW solution and its exact collision term at t = 6.5
    const double t = 6.5, K = 1 - std::exp(-t / 6), dK = std::exp(-t / 6) / 6;
    const size_t G = static_cast<size_t>(Nv) * Nv * Nv;
    std::vector<double> f_h(G), Q_exact(G), Q_h(G);
    const double norm = 1 / (2 * std::pow(2 * pi * K, 1.5));
#pragma omp parallel for collapse(2)
    for (int i = 0; i < Nv; ++i)
        for (int j = 0; j < Nv; ++j)
            for (int k = 0; k < Nv; ++k) {
                const size_t idx = (static_cast<size_t>(i) * Nv + j) * Nv + k;
                const double r2 = v[i] * v[i] + v[j] * v[j] + v[k] * v[k];
                const double gauss = std::exp(-r2 / (2 * K));
                const double fv = norm * gauss * ((5 * K - 3) / K + (1 - K) / (K * K) * r2);
                double q = (-3 / (2 * K) + r2 / (2 * K * K)) * fv;
                q += norm * gauss * (3 / (K * K) + (K - 2) / (K * K * K) * r2);
                f_h[idx] = fv;
                Q_exact[idx] = q * dK;
            }
    const bool random_input = a.input == "random";
    if (random_input)
        for (size_t i = 0; i < G; ++i) f_h[i] *= 1.0 + 0.1 * unit_random(i, 0x5EED);

    auto gl_quadrature = std::make_shared<GaussLegendreQuadrature>(a.Ngl, 0, R);
    auto spherical_quadrature = std::make_shared<SphericalDesign>(a.Ns);

    HIP_OR_DIE(hipSetDevice(a.device));
    double *f_d = nullptr, *Q_d = nullptr;
    HIP_OR_DIE(hipMalloc(reinterpret_cast<void**>(&f_d), G * sizeof(double)));
    HIP_OR_DIE(hipMalloc(reinterpret_cast<void**>(&Q_d), G * sizeof(double)));
    HIP_OR_DIE(hipMemcpy(f_d, f_h.data(), G * sizeof(double), hipMemcpyHostToDevice));

    BoltzmannOperator<HIP_Backend> collision_operator(gl_quadrature, spherical_quadrature, Nv, Nv, Nv, gamma, b_gamma, L);
    collision_operator.setPrecision(a.precision);
    collision_operator.setDevice(a.device);
    collision_operator.setExactReductions(a.exact, a.hermitian);

    using clk = std::chrono::steady_clock;
    const auto t_init = clk::now();
    collision_operator.initialize();
    std::cout << "Initialization time (s): " << std::chrono::duration<double>(clk::now() - t_init).count() << " seconds\n";

    for (int w = 0; w < a.warmup; ++w) collision_operator(Q_d, f_d);
    std::vector<double> collision_times;
    collision_times.reserve(a.trials);
    for (int trial = 0; trial < a.trials; ++trial) {
        const auto t0 = clk::now();
        collision_operator(Q_d, f_d);                      // blocking, like the reference (cu:146-149)
        collision_times.push_back(std::chrono::duration<double>(clk::now() - t0).count());
    }
    print_stats_summary(collision_operator.getBackendName(), collision_times);

    HIP_OR_DIE(hipMemcpy(Q_h.data(), Q_d, G * sizeof(double), hipMemcpyDeviceToHost));

    if (!random_input) {
        double err_L1 = 0, err_L2 = 0, err_Linf = 0;
#pragma omp parallel for reduction(+ : err_L1, err_L2) reduction(max : err_Linf)
        for (size_t i = 0; i < G; ++i) {
            const double d = std::abs(Q_h[i] - Q_exact[i]);
            err_L1 += d;
            err_L2 += d * d;
            err_Linf = std::max(err_Linf, d);
        }
        err_L1 *= dv * dv * dv;
        err_L2 = std::sqrt(err_L2 * dv * dv * dv);
        std::cout << "Approximation errors:\n";
        std::cout << "L1 error: " << err_L1 << "\n";
        std::cout << "L2 error: " << err_L2 << "\n";
        std::cout << "Linf error: " << err_Linf << "\n\n";
    } else {
        double s = 0;
        for (size_t i = 0; i < G; ++i) s += std::abs(Q_h[i]);
        std::cout << "sum |Q| = " << std::scientific << std::setprecision(10) << s << "\n\n";
    }

    // one machine-readable line: throughput and algorithmic bandwidth ((6B + 9) G c bytes / eval)
    const RunStats st = summarize(collision_times);
    const double B = static_cast<double>(a.Ngl) * a.Ns, c = a.precision == 64 ? 16.0 : 8.0;
    const double bytes = (6 * B + 9) * static_cast<double>(G) * c;
    std::cout << std::defaultfloat << std::setprecision(6) << "{\"backend\": \"" << collision_operator.getBackendName()
              << "\", \"Nv\": " << Nv << ", \"Ngl\": " << a.Ngl << ", \"Ns\": " << a.Ns
              << ", \"evals_per_s\": " << 1.0 / st.mean << ", \"alg_GBps\": " << bytes / st.mean / 1e9
              << ", \"frac_hbm_peak_8TBps\": " << bytes / st.mean / 8.0e12 << "}\n";

    HIP_OR_DIE(hipFree(f_d));
    HIP_OR_DIE(hipFree(Q_d));
    return 0;
}
