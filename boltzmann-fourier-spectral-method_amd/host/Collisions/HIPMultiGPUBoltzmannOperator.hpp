// BoltzmannOperator<HIP_MultiGPU_Backend> -- the same operator interface (reference:
// Collisions/AbstractCollisionOperator.hpp:7-26, constructor of Collisions/CUDABoltzmannOperator.hpp:48-54) spread over
// several MI355X GPUs of one node from ONE process, so that a driver written against the reference's API uses all
// GPUs without any change besides the backend tag:
//
//     BoltzmannOperator<HIP_MultiGPU_Backend> collision_operator(gl, sph, Nv, Nv, Nv, gamma, b_gamma, L);
//     collision_operator.setDevices({0, 1, 2, 3, 4, 5, 6, 7});      // optional: default = every visible device
//     collision_operator.initialize();
//     collision_operator(Q, f);          // device pointers on the FIRST device of the list, blocking
//
// New functionality (the reference is single-device).  Per evaluation: f is broadcast from the first device, every
// device evaluates its contiguous shard of the M_gl * M_sph quadrature directions and inverse-transforms its own partial
// sum (bfsm_collide_partial_async; the first device also subtracts the loss term), and ONE grouped RCCL reduce over
// xGMI sums the real Q into the caller's buffer.  With one device no RCCL call is made.
// Each device is driven by its own host thread (created by initialize()), so the devices' launch sequences and their
// two RCCL calls are issued concurrently; the calling thread only publishes (Q, f) and waits.  f must be complete, or
// enqueued on the first device's legacy default stream (what the reference's driver does), when the call is made; a
// producer on any other stream is named with setInputStream().
// No HIP or RCCL type appears in this header; the implementation is host/HIPMultiGPUBoltzmannOperator.cpp.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "AbstractCollisionOperator.hpp"
#include "BoltzmannOperator.hpp"
#include "../Quadratures/GaussLegendre.hpp"
#include "../Quadratures/SphericalDesign.hpp"
#include "bfsm.h"

struct HIP_MultiGPU_Backend {};

template <>
class BoltzmannOperator<HIP_MultiGPU_Backend> : public AbstractCollisionOperator {
public:
    BoltzmannOperator(std::shared_ptr<GaussLegendreQuadrature> gl_quadrature,
                      std::shared_ptr<SphericalQuadrature> spherical_quadrature,
                      int Nvx, int Nvy, int Nvz, double gamma, double b_gamma, double L);
    ~BoltzmannOperator() override;
    BoltzmannOperator(const BoltzmannOperator&) = delete;
    BoltzmannOperator& operator=(const BoltzmannOperator&) = delete;

    // Knobs, to be set before initialize().
    void setDevices(const std::vector<int>& device_ordinals);   // first entry = the device that owns Q and f
    void setPrecision(int bits);                                // 64 (default) or 32
    void setExactReductions(bool on, bool hermitian = false);   // opt-in exact work reductions (include/bfsm.h)
    void setForceCollectives(bool on);                          // use RCCL even with a single device (tests)
    void setMaxChunk(int n);                                    // directions resident at once per device (0: default)
    void setMaxBatch(int n);                                    // distributions per computeCollisionBatch call
    void setProfiling(bool on);                                 // per-kernel events on every device (counters())
    void setTimeoutSeconds(double s);                           // watchdog of a blocking call (default 300 s; 0: off)
    // The stream ON THE FIRST DEVICE on which the caller produces f (may be changed between calls).  Default: none --
    // the operator's stream on that device is a blocking stream, i.e. ordered behind the legacy default stream only (what
    // the reference's driver uses); f produced on any other (non-blocking) stream must be complete when the call is made,
    // or be named here: the broadcast then waits for an event recorded on it at the call.
    void setInputStream(void* hip_stream);
    void clearInputStream();

    void initialize() override;
    std::string getBackendName() const override { return "HIP"; }
    void computeCollision(double* Q, const double* f_in) override;      // device pointers on devices()[0], blocking
    void operator()(double* Q, const double* f_in) override { computeCollision(Q, f_in); }

    // n_batch <= setMaxBatch() distributions, [n_batch][Nvx*Nvy*Nvz] device arrays on devices()[0]: one broadcast, one
    // batched shard evaluation per device, ONE reduce
    void computeCollisionBatch(double* Q, const double* f_in, int n_batch);

    const std::vector<int>& devices() const;
    bfsm_counters counters(int device_index) const;             // of the device_index-th device of devices()

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};
