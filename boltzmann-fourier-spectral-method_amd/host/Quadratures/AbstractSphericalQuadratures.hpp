// Quadrature rule on the unit sphere: Cartesian node coordinates + weights
// (interface of the reference's Quadratures/AbstractSphericalQuadratures.hpp:21-42).
#pragma once
#include <iostream>
#include <vector>

class SphericalQuadrature {
public:
    virtual ~SphericalQuadrature() = default;
    const std::vector<double>& getWeights() const { return weights; }
    const std::vector<double>& getx() const { return x; }
    const std::vector<double>& gety() const { return y; }
    const std::vector<double>& getz() const { return z; }
    int getNumberOfPoints() const { return static_cast<int>(weights.size()); }
    void printQuadratureInfo() const {
        for (std::size_t i = 0; i < x.size(); ++i)
            std::cout << "w=" << weights[i] << " (" << x[i] << ", " << y[i] << ", " << z[i] << ")\n";
    }

protected:
    std::vector<double> weights, x, y, z;
};
