// Backend-agnostic collision-operator interface.
// Same contract as the reference's Collisions/AbstractCollisionOperator.hpp:7-26 (initialize / getBackendName /
// computeCollision / operator()), so a maxwell_bkw_* style driver can hold any backend through this base class.
#pragma once
#include <string>

class AbstractCollisionOperator {
public:
    AbstractCollisionOperator() = default;
    virtual ~AbstractCollisionOperator() = default;

    // Allocate scratch, build plans / tables.  Constructors do no work; all set-up happens here.
    virtual void initialize() = 0;

    // Short backend tag ("HIP", "FFTW", "CUDA", ...).
    virtual std::string getBackendName() const = 0;

    // Q = Q(f_in, f_in) on the N^3 velocity grid.  Memory space of the two pointers is backend-defined:
    // host for CPU backends, device for GPU backends (reference: maxwell_bkw_cuda.cu:119-126,147).
    virtual void computeCollision(double* Q, const double* f_in) = 0;

    // Function-call sugar for computeCollision.
    virtual void operator()(double* Q, const double* f_in) = 0;
};
