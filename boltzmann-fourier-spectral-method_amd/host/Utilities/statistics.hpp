// Run-time statistics in the reference's report format (Utilities/statistics.hpp:53-63) so logs diff against
// the archived Results/*.txt: mean / min / max / sample standard deviation (n-1 denominator, hpp:49).
#pragma once
#include <cmath>
#include <cstddef>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

struct RunStats {
    double mean = 0, lo = 0, hi = 0, stdev = 0;
    std::size_t n = 0;
};

inline RunStats summarize(const std::vector<double>& samples) {
    RunStats s;
    s.n = samples.size();
    if (s.n == 0) return s;
    s.lo = s.hi = samples.front();
    double sum = 0;
    for (double t : samples) {
        sum += t;
        if (t < s.lo) s.lo = t;
        if (t > s.hi) s.hi = t;
    }
    s.mean = sum / static_cast<double>(s.n);
    double ss = 0;
    for (double t : samples) ss += (t - s.mean) * (t - s.mean);
    s.stdev = std::sqrt(ss / static_cast<double>(s.n > 1 ? s.n - 1 : 1));
    return s;
}

inline void print_stats_summary(const std::string& device_name, const std::vector<double>& samples) {
    const RunStats s = summarize(samples);
    std::cout << "\nRun statistics for " << device_name << "\n"
              << "Total number of samples taken: " << s.n << "\n"
              << std::scientific << std::setprecision(8)
              << "Mean runtime (s): " << s.mean << "\n"
              << "Min runtime (s): " << s.lo << "\n"
              << "Max runtime (s): " << s.hi << "\n"
              << "stdev: " << s.stdev << "\n\n";
    // std::cout stays in scientific / precision 8, exactly as the reference leaves it (statistics.hpp:56-59): the
    // drivers' error norms that follow therefore print 9 significant digits, like the archived Results/*.txt
}
