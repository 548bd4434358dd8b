// BoltzmannOperator<HIP_Backend> -- the MI355X drop-in for BoltzmannOperator<CUDA_Backend>
// (reference: Collisions/CUDABoltzmannOperator.hpp:44-131).  Same constructor signature, same life-cycle
// (construct -> initialize() -> operator()(Q, f) with DEVICE pointers, blocking), same error behaviour
// (message on std::cerr, then std::exit(EXIT_FAILURE), reference hpp:20-38).
//
// It is a thin C++ face over the C-ABI of include/bfsm.h: no HIP types appear here, all device work lives in
// libbfsm_hip.so.
#pragma once
#include <memory>
#include <string>

#include "AbstractCollisionOperator.hpp"
#include "BoltzmannOperator.hpp"
#include "../Quadratures/GaussLegendre.hpp"
#include "../Quadratures/SphericalDesign.hpp"
#include "bfsm.h"

struct HIP_Backend {};

template <>
class BoltzmannOperator<HIP_Backend> : public AbstractCollisionOperator {
public:
    BoltzmannOperator(std::shared_ptr<GaussLegendreQuadrature> gl_quadrature,
                      std::shared_ptr<SphericalQuadrature> spherical_quadrature,
                      int Nvx, int Nvy, int Nvz, double gamma, double b_gamma, double L);
    ~BoltzmannOperator() override;

    BoltzmannOperator(const BoltzmannOperator&) = delete;
    BoltzmannOperator& operator=(const BoltzmannOperator&) = delete;

    // Optional knobs, to be set before initialize() (the FFTW backend has setWisdomFileName in the same role,
    // reference FFTWBoltzmannOperator.hpp:39-41).
    void setPrecision(int bits) { precision_ = bits; }                 // 64 (default) or 32
    void setDevice(int ordinal) { device_ = ordinal; }
    void setDirectionShard(long long begin, long long end) { dir_begin_ = begin; dir_end_ = end; }
    void setMaxChunk(int n) { max_chunk_ = n; }
    void setProfiling(bool on) { flags_ = on ? (flags_ | BFSM_FLAG_PROFILE) : (flags_ & ~BFSM_FLAG_PROFILE); }
    // Opt-in exact work reductions (include/bfsm.h: BFSM_FLAG_EXACT_REDUCTIONS; hermitian adds BFSM_FLAG_HERMITIAN).
    void setExactReductions(bool on, bool hermitian = false) {
        flags_ &= ~(BFSM_FLAG_EXACT_REDUCTIONS | BFSM_FLAG_HERMITIAN);
        if (on) flags_ |= BFSM_FLAG_EXACT_REDUCTIONS | (hermitian ? BFSM_FLAG_HERMITIAN : 0);
    }
    void setMaxBatch(int n) { max_batch_ = n; }

    void initialize() override;
    std::string getBackendName() const override { return bfsm_backend_name(); }
    void computeCollision(double* Q, const double* f_in) override;      // device pointers, blocking
    void operator()(double* Q, const double* f_in) override { computeCollision(Q, f_in); }

    // Batch of n_batch <= setMaxBatch() distributions, [n_batch][Nvx*Nvy*Nvz] device arrays, one set of launches.
    void computeCollisionBatch(double* Q, const double* f_in, int n_batch);
    // Batch x direction shard: partial results of every member [with the loss term]; the caller sums Q over the ranks.
    void collideBatchPartial(double* Q, const double* f_in, int n_batch, bool with_loss, void* stream = nullptr);

    // Sharded evaluation (multi-GPU): partial gain -> caller's RCCL reduce on qhatBuffer() -> finish.
    void gainPartial(const double* f_in, void* stream = nullptr);
    void finish(double* Q, const double* f_in, void* stream = nullptr);
    void finishPartial(double* Q, const double* f_in, bool with_loss, void* stream = nullptr);
    // gainPartial + finishPartial as one call (slab reduce fused into the tail; qhatBuffer() is not updated)
    void collidePartial(double* Q, const double* f_in, bool with_loss, void* stream = nullptr);
    // The same call returning the C-ABI status instead of print-and-exit (for callers that own the failure policy, e.g.
    // the device threads of the multi-GPU operator); lastError() is the message of the last non-zero status.
    int collidePartialStatus(double* Q, const double* f_in, bool with_loss, void* stream = nullptr) noexcept;
    int collideBatchPartialStatus(double* Q, const double* f_in, int n_batch, bool with_loss, void* stream = nullptr) noexcept;
    const char* lastError() const noexcept;
    void* qhatBuffer(size_t* n_elems, int* precision) const;
    void synchronize();
    bfsm_counters counters() const;
    bfsm_handle handle() const { return handle_; }

protected:
    const int Nvx, Nvy, Nvz;
    const double gamma, b_gamma, L;
    const std::shared_ptr<GaussLegendreQuadrature> gl_quadrature;
    const std::shared_ptr<SphericalQuadrature> spherical_quadrature;

private:
    void check(int rc, const char* what) const;
    bfsm_handle handle_ = nullptr;
    int precision_ = BFSM_F64;
    int device_ = 0;
    long long dir_begin_ = 0, dir_end_ = 0;
    int max_chunk_ = 0;
    int flags_ = BFSM_FLAG_NONE;
    int max_batch_ = 0;
};
