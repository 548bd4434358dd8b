#pragma once
// pi to the digits the reference uses (Utilities/constants.hpp:7) so that derived constants agree bit for bit.
constexpr double pi = 3.14159265358979323846;
