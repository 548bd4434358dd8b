// Gauss-Legendre rule on [a,b] with the semantics the reference obtains from GSL's glfixed table
// (Quadratures/GaussLegendre.hpp:10-24): ascending nodes x_i = (a+b)/2 + (b-a)/2 t_i, weights (b-a)/2 w_i.
// GSL is not a dependency here: roots of P_n by Newton iteration in long double.
#pragma once
#include <cmath>
#include <stdexcept>

#include "AbstractQuadrature.hpp"

class GaussLegendreQuadrature : public AbstractQuadrature {
public:
    GaussLegendreQuadrature(int n_points, double a, double b) {
        if (n_points < 1) throw std::invalid_argument("Gauss-Legendre rule needs at least one point");
        const int n = n_points;
        nodes.assign(n, 0.0);
        weights.assign(n, 0.0);
        const long double PI_L = 3.141592653589793238462643383279502884L;
        const long double half = (static_cast<long double>(b) - a) / 2, mid = (static_cast<long double>(a) + b) / 2;
        auto legendre = [n](long double x, long double& pn, long double& dpn) {
            long double p0 = 1, p1 = x;
            for (int k = 2; k <= n; ++k) {
                const long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
                p0 = p1;
                p1 = pk;
            }
            pn = p1;
            dpn = n * (x * p1 - p0) / (x * x - 1);
        };
        for (int i = 0; i < (n + 1) / 2; ++i) {
            long double x = std::cos(PI_L * (i + 0.75L) / (n + 0.5L));   // i-th largest root
            long double pn, dpn;
            for (int it = 0; it < 100; ++it) {
                legendre(x, pn, dpn);
                const long double dx = pn / dpn;
                x -= dx;
                if (std::fabs(dx) < 1e-19L) break;
            }
            legendre(x, pn, dpn);
            const long double w = 2 / ((1 - x * x) * dpn * dpn);
            nodes[n - 1 - i] = static_cast<double>(mid + half * x);
            nodes[i] = static_cast<double>(mid - half * x);
            weights[n - 1 - i] = weights[i] = static_cast<double>(half * w);
        }
        if (n % 2 == 1) nodes[n / 2] = static_cast<double>(mid);
    }

    void printQuadratureInfo() const override {
        std::cout << "Gauss-Legendre Quadrature:\n";
        AbstractQuadrature::printQuadratureInfo();
    }
};
