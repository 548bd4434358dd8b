// CPU-only checks of the C++ mirror of the reference's host interface (no GPU, no libbfsm_hip.so needed):
// quadrature providers, their error behaviour (reference: SphericalDesign.cpp:7-9,22-31; GaussLegendre.hpp:15-17),
// the statistics printer's format (Utilities/statistics.hpp:53-63) and the abstract-operator plumbing.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>

#include "Collisions/AbstractCollisionOperator.hpp"
#include "Collisions/BoltzmannOperator.hpp"
#include "Collisions/HIPBoltzmannOperator.hpp"            // declarations only: no HIP / RCCL type may leak into the
#include "Collisions/HIPMultiGPUBoltzmannOperator.hpp"    // headers a reference driver includes (plain g++ compiles them)
#include "Quadratures/GaussLegendre.hpp"
#include "Quadratures/SphericalDesign.hpp"
#include "Utilities/constants.hpp"
#include "Utilities/statistics.hpp"

static int failures = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

struct Dummy_Backend {};
template <>
class BoltzmannOperator<Dummy_Backend> : public AbstractCollisionOperator {   // tag dispatch works like the reference's
public:
    void initialize() override { ready = true; }
    std::string getBackendName() const override { return "Dummy"; }
    void computeCollision(double* Q, const double* f) override { Q[0] = 2 * f[0]; }
    void operator()(double* Q, const double* f) override { computeCollision(Q, f); }
    bool ready = false;
};

int main(int argc, char** argv) {
    const std::string data_dir = argc > 1 ? argv[1] : "";

    // Gauss-Legendre: exactness degree 2n-1, ascending nodes, GSL glfixed semantics on [a, b]
    for (int n : {1, 2, 5, 8, 16, 31, 64}) {
        GaussLegendreQuadrature gl(n, 0.0, 10.0);
        CHECK(gl.getNumberOfPoints() == n);
        const auto& x = gl.getNodes();
        const auto& w = gl.getWeights();
        double sum = 0, m1 = 0, mk = 0;
        const int k = 2 * n - 1;
        for (int i = 0; i < n; ++i) {
            if (i) CHECK(x[i] > x[i - 1]);
            CHECK(x[i] > 0.0 && x[i] < 10.0 && w[i] > 0.0);
            sum += w[i];
            m1 += w[i] * x[i];
            mk += w[i] * std::pow(x[i] / 10.0, k);
        }
        CHECK(std::fabs(sum - 10.0) < 1e-13);
        if (n >= 1) CHECK(std::fabs(m1 - 50.0) < 1e-12);
        CHECK(std::fabs(mk - 10.0 / (k + 1)) < 1e-13);     // integral of (x/10)^k over [0,10]
    }
    {   // two-point rule on [-1,1]: +-1/sqrt(3), weights 1
        GaussLegendreQuadrature g2(2, -1.0, 1.0);
        CHECK(std::fabs(g2.getNodes()[0] + 1 / std::sqrt(3.0)) < 1e-15 && std::fabs(g2.getNodes()[1] - 1 / std::sqrt(3.0)) < 1e-15);
        CHECK(std::fabs(g2.getWeights()[0] - 1.0) < 1e-15);
    }
    bool threw = false;
    try { GaussLegendreQuadrature bad(0, 0.0, 1.0); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);

    // Spherical designs: all nine shipped sizes, unit vectors, antipodal pairing, weights 4 pi / N, quadrature of
    // low-degree polynomials (a t-design integrates degree <= t exactly: check <x^2> = 1/3, <x y> = 0, <z> = 0)
    const int sizes[] = {6, 12, 32, 48, 70, 94, 120, 156, 192};
    for (int N : sizes) {
        SphericalDesign sd(N, data_dir);
        CHECK(sd.getNumberOfPoints() == N);
        double sxx = 0, sxy = 0, sz = 0, wsum = 0;
        for (int s = 0; s < N; ++s) {
            const double x = sd.getx()[s], y = sd.gety()[s], z = sd.getz()[s], w = sd.getWeights()[s];
            CHECK(std::fabs(x * x + y * y + z * z - 1) < 1e-15);
            CHECK(w == (4 * pi) / N);
            const int t = (s + N / 2) % N;
            CHECK(sd.getx()[t] == -x && sd.gety()[t] == -y && sd.getz()[t] == -z);
            sxx += w * x * x; sxy += w * x * y; sz += w * z; wsum += w;
        }
        CHECK(std::fabs(wsum - 4 * pi) < 1e-13);
        CHECK(std::fabs(sxx - 4 * pi / 3) < 1e-13 && std::fabs(sxy) < 1e-13 && std::fabs(sz) < 1e-13);
    }
    threw = false;
    try { SphericalDesign bad(13, data_dir); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { SphericalDesign bad(-1, data_dir); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { SphericalDesign bad(12, "/nonexistent/dir"); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
    if (argc > 2) {   // a directory holding the same designs in the reference's table format (ssTTT.NNN.txt)
        for (int N : {12, 48}) {
            SphericalDesign a(N, data_dir), b(N, argv[2]);
            for (int s = 0; s < N; ++s)
                CHECK(a.getx()[s] == b.getx()[s] && a.gety()[s] == b.gety()[s] && a.getz()[s] == b.getz()[s] &&
                      a.getWeights()[s] == b.getWeights()[s]);
        }
    }
    SphericalDesign::setDataDirectory(data_dir);
    CHECK(SphericalDesign(48).getNumberOfPoints() == 48);      // default-directory constructor honours the setter

    // statistics: n-1 denominator, report format diff-able against the reference's Results/*.txt
    {
        const RunStats s = summarize({1.0, 2.0, 3.0, 4.0});
        CHECK(s.n == 4 && s.mean == 2.5 && s.lo == 1.0 && s.hi == 4.0);
        CHECK(std::fabs(s.stdev - std::sqrt(5.0 / 3.0)) < 1e-15);
        CHECK(summarize({7.0}).stdev == 0.0);
        std::ostringstream cap;
        auto* old = std::cout.rdbuf(cap.rdbuf());
        print_stats_summary("HIP", {0.5, 0.5});
        std::cout.rdbuf(old);
        const std::string out = cap.str();
        CHECK(out.find("Run statistics for HIP\n") != std::string::npos);
        CHECK(out.find("Total number of samples taken: 2\n") != std::string::npos);
        CHECK(out.find("Mean runtime (s): 5.00000000e-01\n") != std::string::npos);
        CHECK(out.find("stdev: 0.00000000e+00\n") != std::string::npos);
    }

    // operator interface: polymorphic use through the abstract base, like the reference's drivers
    {
        std::unique_ptr<AbstractCollisionOperator> op(new BoltzmannOperator<Dummy_Backend>());
        op->initialize();
        double f = 21, Q = 0;
        (*op)(&Q, &f);
        CHECK(Q == 42 && op->getBackendName() == "Dummy");
    }
    std::printf(failures ? "%d check(s) failed\n" : "all host-mirror checks passed\n", failures);
    return failures ? 1 : 0;
}
