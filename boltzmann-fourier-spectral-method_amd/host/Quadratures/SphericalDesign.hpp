// N-point symmetric spherical t-design with equal weights 4*pi/N (reference: Quadratures/SphericalDesign.hpp:22,
// SphericalDesign.cpp:6-50).  N must be one of 6, 12, 32, 48, 70, 94, 120, 156, 192.
// The tables are read from a data directory that is configurable (setDataDirectory, or the BFSM_DESIGN_DIR
// environment variable, or the build-time default) instead of the reference's hard-coded absolute path (cpp:13-21).
#pragma once
#include <string>

#include "AbstractSphericalQuadratures.hpp"

class SphericalDesign : public SphericalQuadrature {
public:
    explicit SphericalDesign(int N);
    SphericalDesign(int N, const std::string& data_dir);
    static void setDataDirectory(const std::string& dir);
    static std::string dataDirectory();

private:
    void load(int N, const std::string& dir);
};
