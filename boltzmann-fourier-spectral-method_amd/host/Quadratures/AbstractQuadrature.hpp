// 1-D quadrature rule: nodes + weights (interface of the reference's Quadratures/AbstractQuadrature.hpp:17-29).
#pragma once
#include <iostream>
#include <vector>

class AbstractQuadrature {
public:
    virtual ~AbstractQuadrature() = default;
    const std::vector<double>& getWeights() const { return weights; }
    const std::vector<double>& getNodes() const { return nodes; }
    int getNumberOfPoints() const { return static_cast<int>(weights.size()); }
    virtual void printQuadratureInfo() const {
        std::cout << "Quadrature Weights:";
        for (double w : weights) std::cout << ' ' << w;
        std::cout << "\nQuadrature Nodes:";
        for (double x : nodes) std::cout << ' ' << x;
        std::cout << std::endl;
    }

protected:
    std::vector<double> weights, nodes;
};
