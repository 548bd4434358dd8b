// fft_benchmark_hip -- batched 3-D complex transforms with the operator's own hand-written FFT kernels.
//
// Counterpart of the reference's cufft_benchmark.cu (batch of Ns*Nv transforms of an all-ones Nv^3 array,
// cu:91-117; timed forward transform over `trials`, cu:139-152; scale by 1/Nv^3, inverse, round-trip L1 error
// times dv^3, cu:154-191; "Approximation error" + print_stats_summary report, cu:193-196).  Same flags --Nv --Ns
// -t/--trials.  The transforms go through the C-ABI entry point bfsm_fft3d (in place; the forward result is in the
// library's spectral layout [lx][lz][ly], which the inverse consumes), so this measures the building block of the
// collision operator, not a library FFT.  Extra flags: --precision {64,32}, --device.  Also prints the bandwidth the
// two passes of a 3-D transform reach: 4 * G * sizeof(complex) bytes per transform (each pass reads and writes once).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "Utilities/statistics.hpp"
#include "bfsm.h"

#define HIP_OR_DIE(call)                                                                                          \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            std::cerr << "HIP Error: " << hipGetErrorString(e_) << " at " << __FILE__ << ":" << __LINE__ << std::endl; \
            std::exit(EXIT_FAILURE);                                                                              \
        }                                                                                                         \
    } while (0)

static void bfsm_or_die(int rc, bfsm_handle h, const char* what) {
    if (rc == BFSM_OK) return;
    std::cerr << "HIP backend error in " << what << ": " << bfsm_last_error(h) << std::endl;
    std::exit(EXIT_FAILURE);
}

int main(int argc, char** argv) {
    int Nv = 32, Ns = 32, trials = 1, precision = 64, device = 0;
    for (int i = 1; i < argc; ++i) {
        auto value = [&](const char* name) -> const char* {
            if (std::strcmp(argv[i], name) != 0) return nullptr;
            if (i + 1 >= argc) { std::cerr << "error: missing value for " << name << "\n"; std::exit(EXIT_FAILURE); }
            return argv[++i];
        };
        if (const char* v = value("--Nv")) Nv = std::atoi(v);
        else if (const char* v = value("--Ns")) Ns = std::atoi(v);
        else if (const char* v = value("-t")) trials = std::atoi(v);
        else if (const char* v = value("--trials")) trials = std::atoi(v);
        else if (const char* v = value("--precision")) precision = std::atoi(v);
        else if (const char* v = value("--device")) device = std::atoi(v);
        else { std::cerr << "error: unknown argument " << argv[i] << "\n"; return EXIT_FAILURE; }
    }
    std::cout << "\nRun arguments:\n" << "Nv = " << Nv << "\n" << "Ns = " << Ns << "\n" << "trials = " << trials << "\n";
    if (trials < 1 || Ns < 1) { std::cerr << "error: trials and Ns must be positive\n"; return EXIT_FAILURE; }

    const double S = 5, L = ((3 + std::sqrt(2.0)) / 2) * S, dv = 2 * L / Nv;
    const size_t grid_size = (size_t)Nv * Nv * Nv;
    const int batch_size = Ns * Nv;                       // as the reference sizes its experiment (cu:92)
    const size_t elem = precision == 64 ? 2 * sizeof(double) : 2 * sizeof(float);

    // a handle only carries the grid size / precision here; the quadrature is a one-direction placeholder
    const double one = 1.0, zero = 0.0, wsph = 12.566370614359172;
    bfsm_desc d{};
    d.nvx = d.nvy = d.nvz = Nv;
    d.n_gl = 1; d.n_sph = 1;
    d.gl_nodes = &one; d.gl_wts = &one; d.sph_wts = &wsph; d.sx = &zero; d.sy = &zero; d.sz = &one;
    d.gamma = 0; d.b_gamma = 1; d.L = L; d.precision = precision == 64 ? BFSM_F64 : BFSM_F32; d.device = device;
    bfsm_handle h = nullptr;
    bfsm_or_die(bfsm_create(&d, &h), nullptr, "bfsm_create");
    HIP_OR_DIE(hipSetDevice(device));

    // all-ones input, replicated over the batch (cu:96-117)
    std::vector<unsigned char> ones(grid_size * elem);
    for (size_t i = 0; i < grid_size; ++i) {
        if (precision == 64) { reinterpret_cast<double*>(ones.data())[2 * i] = 1; reinterpret_cast<double*>(ones.data())[2 * i + 1] = 0; }
        else { reinterpret_cast<float*>(ones.data())[2 * i] = 1; reinterpret_cast<float*>(ones.data())[2 * i + 1] = 0; }
    }
    unsigned char* f = nullptr;
    HIP_OR_DIE(hipMalloc((void**)&f, (size_t)batch_size * grid_size * elem));
    auto fill = [&]() {
        for (int b = 0; b < batch_size; ++b)
            HIP_OR_DIE(hipMemcpy(f + (size_t)b * grid_size * elem, ones.data(), grid_size * elem, hipMemcpyHostToDevice));
    };
    fill();

    // timed forward transforms (cu:139-152).  In place: every trial restarts from the all-ones batch.
    using clk = std::chrono::steady_clock;
    std::vector<double> times;
    times.reserve(trials);
    for (int t = 0; t < trials; ++t) {
        if (t) fill();
        HIP_OR_DIE(hipDeviceSynchronize());
        const auto t0 = clk::now();
        bfsm_or_die(bfsm_fft3d(h, f, batch_size, -1), h, "bfsm_fft3d(forward)");   // blocking
        times.push_back(std::chrono::duration<double>(clk::now() - t0).count());
    }
    // inverse, then normalise on the host while comparing (cu:154-191)
    bfsm_or_die(bfsm_fft3d(h, f, batch_size, +1), h, "bfsm_fft3d(inverse)");
    const double scale = 1.0 / (double)grid_size;
    double l1 = 0;
    std::vector<unsigned char> back(grid_size * elem);
    for (int b = 0; b < batch_size; ++b) {
        HIP_OR_DIE(hipMemcpy(back.data(), f + (size_t)b * grid_size * elem, grid_size * elem, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < grid_size; ++i) {
            const double re = precision == 64 ? reinterpret_cast<const double*>(back.data())[2 * i]
                                              : (double)reinterpret_cast<const float*>(back.data())[2 * i];
            l1 += std::abs(1.0 - re * scale);
        }
    }
    l1 *= dv * dv * dv;
    std::cout << "Approximation error (HIP tile + line passes):\n";
    std::cout << "L1 error: " << l1 << "\n";
    print_stats_summary("HIP tile + line passes", times);
    const double best = *std::min_element(times.begin(), times.end());
    std::cout << std::defaultfloat << std::setprecision(6) << "Batch of " << batch_size << " transforms, " << (4.0 * grid_size * elem * batch_size) / 1e9
              << " GB over the two passes: " << (4.0 * grid_size * elem * batch_size) / best / 1e12 << " TB/s (best trial)\n";

    HIP_OR_DIE(hipFree(f));
    bfsm_destroy(h);
    return 0;
}
