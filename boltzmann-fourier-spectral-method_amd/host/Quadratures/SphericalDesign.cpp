#include "SphericalDesign.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "../Utilities/constants.hpp"

#ifndef BFSM_DEFAULT_DESIGN_DIR
#define BFSM_DEFAULT_DESIGN_DIR "data/sph_design"
#endif

namespace {
std::string g_dir;

// degree of the shipped design with N points; 0 if there is none (same set as the reference's switch, cpp:12-24)
int degree_for(int N) {
    switch (N) {
        case 6: return 3;   case 12: return 5;   case 32: return 7;    case 48: return 9;    case 70: return 11;
        case 94: return 13; case 120: return 15; case 156: return 17;  case 192: return 19;
        default: return 0;
    }
}
}  // namespace

void SphericalDesign::setDataDirectory(const std::string& dir) { g_dir = dir; }

std::string SphericalDesign::dataDirectory() {
    if (!g_dir.empty()) return g_dir;
    if (const char* env = std::getenv("BFSM_DESIGN_DIR")) return env;
    return BFSM_DEFAULT_DESIGN_DIR;
}

SphericalDesign::SphericalDesign(int N) { load(N, dataDirectory()); }
SphericalDesign::SphericalDesign(int N, const std::string& data_dir) { load(N, data_dir); }

void SphericalDesign::load(int N, const std::string& dir) {
    if (N <= 0) throw std::invalid_argument("Number of points N must be a positive integer");
    const int t = degree_for(N);
    if (t == 0) throw std::invalid_argument("Invalid value of N");
    // This package's own tables ("sym_design_tTTT_nNNN.dat": comments, a "t n" header, then x y z), or -- so that the
    // directory a user of the reference already has can be used as it is -- the reference's table of the same design
    // ("ssTTT.NNN.txt": N rows of x y z, SphericalDesign.cpp:12-24,38-46).
    char name[64];
    std::snprintf(name, sizeof(name), "sym_design_t%03d_n%03d.dat", t, N);
    std::string path = dir + "/" + name;
    std::ifstream in(path);
    bool header_seen = false;
    if (!in.is_open()) {
        std::snprintf(name, sizeof(name), "ss%03d.%03d.txt", t, N);
        const std::string alt = dir + "/" + name;
        in.open(alt);
        if (!in.is_open()) throw std::runtime_error("Could not open file " + path + " (nor " + alt + ")");
        path = alt;
        header_seen = true;            // the reference's tables have no header row
    }
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream row(line);
        if (!header_seen) {            // "t n"
            int ft = 0, fn = 0;
            row >> ft >> fn;
            if (ft != t || fn != N) throw std::runtime_error("Unexpected header in " + path);
            header_seen = true;
            continue;
        }
        double px, py, pz;
        if (!(row >> px >> py >> pz)) throw std::runtime_error("Malformed line in " + path);
        x.push_back(px);
        y.push_back(py);
        z.push_back(pz);
    }
    if (static_cast<int>(x.size()) != N) throw std::runtime_error("Wrong number of points in " + path);
    weights.assign(N, (4 * pi) / N);   // SphericalDesign.cpp:48
}
