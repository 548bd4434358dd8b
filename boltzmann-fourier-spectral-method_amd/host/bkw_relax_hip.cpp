// bkw_relax_hip -- space-homogeneous relaxation d f / d t = Q(f, f) on one MI355X (the natural caller
// of the collision operator; the reference stops at a single evaluation of Q).  SSP-RK3 (Shu-Osher) from the BKW
// state at t0 to t1 with f resident on the device between evaluations; reports the error against the exact BKW
// solution, mass / energy drift and the entropy at both ends.  Flags: --Nv --Ns --Ngl --t0 --t1 --steps
// --exact-reductions --hermitian --design-dir.
#include <hip/hip_runtime.h>

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

#define HIP_OR_DIE(call)                                                                                          \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            std::cerr << "HIP Error: " << hipGetErrorString(e_) << " at " << __FILE__ << ":" << __LINE__ << "\n"; \
            std::exit(EXIT_FAILURE);                                                                              \
        }                                                                                                         \
    } while (0)

// out = a * x + b * y + c * z   (stage combinations of the Runge-Kutta scheme; caller-side glue, not the hot path)
__global__ void combine3(double* out, double a, const double* x, double b, const double* y, double c, const double* z, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + b * y[i] + c * z[i];
}

static void bkw(int Nv, double L, double t, std::vector<double>& f) {
    const double dv = 2 * L / Nv, K = 1 - std::exp(-t / 6), norm = 1 / (2 * std::pow(2 * pi * K, 1.5));
    for (int i = 0; i < Nv; ++i)
        for (int j = 0; j < Nv; ++j)
            for (int k = 0; k < Nv; ++k) {
                const double vx = -L + dv / 2 + i * dv, vy = -L + dv / 2 + j * dv, vz = -L + dv / 2 + k * dv;
                const double r2 = vx * vx + vy * vy + vz * vz;
                f[((size_t)i * Nv + j) * Nv + k] = norm * std::exp(-r2 / (2 * K)) * ((5 * K - 3) / K + (1 - K) / (K * K) * r2);
            }
}

int main(int argc, char** argv) {
    int Nv = 32, Ns = 32, Ngl = 16, steps = 10;
    double t0 = 5.5, t1 = 6.5;
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
        else if ((v = val("--steps"))) steps = std::atoi(v);
        else if ((v = val("--t0"))) t0 = std::atof(v);
        else if ((v = val("--t1"))) t1 = std::atof(v);
        else if ((v = val("--design-dir"))) design_dir = v;
        else if (std::strcmp(argv[i], "--exact-reductions") == 0) exact = true;
        else if (std::strcmp(argv[i], "--hermitian") == 0) exact = hermitian = true;
        else { std::cerr << "error: unknown argument " << argv[i] << "\n"; return EXIT_FAILURE; }
    }
    if (!design_dir.empty()) SphericalDesign::setDataDirectory(design_dir);
    const double gamma = 0, b_gamma = 1 / (4 * pi), S = 5, R = 2 * S, L = ((3 + std::sqrt(2.0)) / 2) * S;
    const double dv = 2 * L / Nv, dt = (t1 - t0) / steps;
    const size_t G = (size_t)Nv * Nv * Nv;
    std::vector<double> f0(G), f_exact(G), f_h(G);
    bkw(Nv, L, t0, f0);
    bkw(Nv, L, t1, f_exact);

    BoltzmannOperator<HIP_Backend> op(std::make_shared<GaussLegendreQuadrature>(Ngl, 0, R),
                                      std::make_shared<SphericalDesign>(Ns), Nv, Nv, Nv, gamma, b_gamma, L);
    op.setExactReductions(exact, hermitian);
    op.initialize();

    double *f, *f1, *f2, *Q;
    for (double** p : {&f, &f1, &f2, &Q}) HIP_OR_DIE(hipMalloc(reinterpret_cast<void**>(p), G * sizeof(double)));
    HIP_OR_DIE(hipMemcpy(f, f0.data(), G * sizeof(double), hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)((G + 255) / 256);
    const auto t_start = std::chrono::steady_clock::now();
    for (int s = 0; s < steps; ++s) {          // everything on the default stream: in-order, no host round trips
        op.gainPartial(f);  op.finish(Q, f);                                                   // Q(f)
        hipLaunchKernelGGL(combine3, blocks, 256, 0, nullptr, f1, 1.0, f, dt, Q, 0.0, Q, G);   // f1 = f + dt Q
        op.gainPartial(f1); op.finish(Q, f1);
        hipLaunchKernelGGL(combine3, blocks, 256, 0, nullptr, f2, 0.75, f, 0.25, f1, 0.25 * dt, Q, G);
        op.gainPartial(f2); op.finish(Q, f2);
        hipLaunchKernelGGL(combine3, blocks, 256, 0, nullptr, f, 1.0 / 3, f, 2.0 / 3, f2, 2.0 / 3 * dt, Q, G);
    }
    HIP_OR_DIE(hipDeviceSynchronize());
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    HIP_OR_DIE(hipMemcpy(f_h.data(), f, G * sizeof(double), hipMemcpyDeviceToHost));

    double l2 = 0, linf = 0, m0 = 0, m1 = 0, e0 = 0, e1 = 0, h0 = 0, h1 = 0;
    for (int i = 0; i < Nv; ++i)
        for (int j = 0; j < Nv; ++j)
            for (int k = 0; k < Nv; ++k) {
                const size_t idx = ((size_t)i * Nv + j) * Nv + k;
                const double vx = -L + dv / 2 + i * dv, vy = -L + dv / 2 + j * dv, vz = -L + dv / 2 + k * dv;
                const double v2 = vx * vx + vy * vy + vz * vz, d = std::abs(f_h[idx] - f_exact[idx]);
                l2 += d * d; linf = std::max(linf, d);
                m0 += f0[idx]; m1 += f_h[idx]; e0 += f0[idx] * v2; e1 += f_h[idx] * v2;
                if (f0[idx] > 0) h0 += f0[idx] * std::log(f0[idx]);
                if (f_h[idx] > 0) h1 += f_h[idx] * std::log(f_h[idx]);
            }
    const double dv3 = dv * dv * dv;
    std::cout << "BKW relaxation t = " << t0 << " -> " << t1 << ", " << steps << " SSP-RK3 steps (" << 3 * steps
              << " collision evaluations) in " << secs << " s\n"
              << "L2 error vs exact BKW: " << std::sqrt(l2 * dv3) << "\nLinf error: " << linf
              << "\nrelative mass drift: " << std::abs(m1 - m0) / m0 << "\nrelative energy drift: " << std::abs(e1 - e0) / e0
              << "\nentropy: " << h0 * dv3 << " -> " << h1 * dv3 << "\n";
    for (double* p : {f, f1, f2, Q}) HIP_OR_DIE(hipFree(p));
    return 0;
}
