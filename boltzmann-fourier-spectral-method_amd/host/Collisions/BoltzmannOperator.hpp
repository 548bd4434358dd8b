// Tag-dispatched Boltzmann operator: only specialisations exist (reference: Collisions/BoltzmannOperator.hpp:7-8).
// This build provides BoltzmannOperator<HIP_Backend> in HIPBoltzmannOperator.hpp.
#pragma once

template <typename Backend>
class BoltzmannOperator;
