#include "Collisions/HIPMultiGPUBoltzmannOperator.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <iostream>

#include "Collisions/HIPBoltzmannOperator.hpp"

// Same observable failure mode as the reference's HANDLE_CUDA_ERROR (CUDABoltzmannOperator.hpp:20-38): message, exit.
#define MG_HIP(call)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            std::cerr << "HIP Error: " << hipGetErrorString(e_) << " at " << __FILE__ << ":" << __LINE__ << "\n"; \
            std::exit(EXIT_FAILURE);                                                                              \
        }                                                                                                         \
    } while (0)
#define MG_RCCL(call)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (call);                                                                                  \
        if (r_ != ncclSuccess) {                                                                                   \
            std::cerr << "RCCL Error: " << ncclGetErrorString(r_) << " at " << __FILE__ << ":" << __LINE__ << "\n"; \
            std::exit(EXIT_FAILURE);                                                                               \
        }                                                                                                          \
    } while (0)

struct BoltzmannOperator<HIP_MultiGPU_Backend>::Impl {
    std::shared_ptr<GaussLegendreQuadrature> gl;
    std::shared_ptr<SphericalQuadrature> sph;
    int Nvx, Nvy, Nvz;
    double gamma, b_gamma, L;
    std::vector<int> devs;                 // empty until setDevices() / initialize()
    int precision = 64;
    bool exact = false, hermitian = false, force_collectives = false;

    bool ready = false, use_rccl = false;
    std::vector<std::unique_ptr<BoltzmannOperator<HIP_Backend>>> ops;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    std::vector<double*> f_rep, Q_rep;     // replicas on devices 1..P-1 (entry 0 unused: the caller's buffers)

    void release() {
        if (!ready) return;
        int prev = 0;
        (void)hipGetDevice(&prev);
        for (size_t g = 0; g < devs.size(); ++g) {
            (void)hipSetDevice(devs[g]);
            ops[g].reset();
            if (f_rep[g]) (void)hipFree(f_rep[g]);
            if (Q_rep[g]) (void)hipFree(Q_rep[g]);
            if (streams[g]) (void)hipStreamDestroy(streams[g]);
            if (use_rccl) ncclCommDestroy(comms[g]);
        }
        (void)hipSetDevice(prev);
        ops.clear(); streams.clear(); comms.clear(); f_rep.clear(); Q_rep.clear();
        ready = false;
    }
};

BoltzmannOperator<HIP_MultiGPU_Backend>::BoltzmannOperator(std::shared_ptr<GaussLegendreQuadrature> gl,
                                                           std::shared_ptr<SphericalQuadrature> sph,
                                                           int nvx, int nvy, int nvz, double gamma, double b_gamma, double L)
    : impl_(new Impl()) {
    impl_->gl = std::move(gl);
    impl_->sph = std::move(sph);
    impl_->Nvx = nvx; impl_->Nvy = nvy; impl_->Nvz = nvz;
    impl_->gamma = gamma; impl_->b_gamma = b_gamma; impl_->L = L;
}

BoltzmannOperator<HIP_MultiGPU_Backend>::~BoltzmannOperator() { impl_->release(); }

void BoltzmannOperator<HIP_MultiGPU_Backend>::setDevices(const std::vector<int>& d) { impl_->devs = d; }
void BoltzmannOperator<HIP_MultiGPU_Backend>::setPrecision(int bits) { impl_->precision = bits; }
void BoltzmannOperator<HIP_MultiGPU_Backend>::setExactReductions(bool on, bool hermitian) {
    impl_->exact = on;
    impl_->hermitian = on && hermitian;
}
void BoltzmannOperator<HIP_MultiGPU_Backend>::setForceCollectives(bool on) { impl_->force_collectives = on; }
const std::vector<int>& BoltzmannOperator<HIP_MultiGPU_Backend>::devices() const { return impl_->devs; }

void BoltzmannOperator<HIP_MultiGPU_Backend>::initialize() {
    Impl& m = *impl_;
    m.release();
    int ndev = 0;
    MG_HIP(hipGetDeviceCount(&ndev));
    if (m.devs.empty())
        for (int g = 0; g < ndev; ++g) m.devs.push_back(g);
    for (size_t g = 0; g < m.devs.size(); ++g) {
        const bool dup = std::count(m.devs.begin(), m.devs.end(), m.devs[g]) != 1;
        if (m.devs[g] < 0 || m.devs[g] >= ndev || dup) {
            std::cerr << "HIP backend error in initialize: device list must name distinct visible devices (" << ndev
                      << " visible)" << std::endl;
            std::exit(EXIT_FAILURE);
        }
    }
    const int P = static_cast<int>(m.devs.size());
    const long long B = static_cast<long long>(m.gl->getNumberOfPoints()) * m.sph->getNumberOfPoints();
    const size_t G = static_cast<size_t>(m.Nvx) * m.Nvy * m.Nvz;
    int prev = 0;
    MG_HIP(hipGetDevice(&prev));
    m.use_rccl = P > 1 || m.force_collectives;
    m.ops.resize(P); m.streams.assign(P, nullptr); m.comms.assign(P, nullptr);
    m.f_rep.assign(P, nullptr); m.Q_rep.assign(P, nullptr);
    m.ready = true;                       // from here on release() has something to undo
    if (m.use_rccl) MG_RCCL(ncclCommInitAll(m.comms.data(), P, m.devs.data()));
    for (int g = 0; g < P; ++g) {
        MG_HIP(hipSetDevice(m.devs[g]));
        MG_HIP(hipStreamCreate(&m.streams[g]));
        if (g > 0) {
            MG_HIP(hipMalloc(reinterpret_cast<void**>(&m.f_rep[g]), G * sizeof(double)));
            MG_HIP(hipMalloc(reinterpret_cast<void**>(&m.Q_rep[g]), G * sizeof(double)));
        }
        m.ops[g].reset(new BoltzmannOperator<HIP_Backend>(m.gl, m.sph, m.Nvx, m.Nvy, m.Nvz, m.gamma, m.b_gamma, m.L));
        m.ops[g]->setDevice(m.devs[g]);
        m.ops[g]->setPrecision(m.precision);
        m.ops[g]->setExactReductions(m.exact, m.hermitian);
        const long long base = B / P, rem = B % P;        // contiguous, balanced shards (== bfsm.shard_range)
        const long long b0 = g * base + std::min<long long>(g, rem), b1 = b0 + base + (g < rem ? 1 : 0);
        m.ops[g]->setDirectionShard(b0, b1);
        m.ops[g]->initialize();
    }
    MG_HIP(hipSetDevice(prev));
}

void BoltzmannOperator<HIP_MultiGPU_Backend>::computeCollision(double* Q, const double* f_in) {
    Impl& m = *impl_;
    if (!m.ready) {
        std::cerr << "HIP backend error in computeCollision: initialize() has not been called" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    const int P = static_cast<int>(m.devs.size());
    const size_t G = static_cast<size_t>(m.Nvx) * m.Nvy * m.Nvz;
    int prev = 0;
    MG_HIP(hipGetDevice(&prev));
    // the call is blocking like the reference's (cu:218), and the caller's earlier work on f (default stream of the
    // first device) must be visible to the private streams used here
    MG_HIP(hipSetDevice(m.devs[0]));
    MG_HIP(hipDeviceSynchronize());
    if (m.use_rccl) {                      // f: first device -> all (in place on the root)
        MG_RCCL(ncclGroupStart());
        for (int g = 0; g < P; ++g) {
            double* buf = g == 0 ? const_cast<double*>(f_in) : m.f_rep[g];
            MG_RCCL(ncclBroadcast(buf, buf, G, ncclDouble, 0, m.comms[g], m.streams[g]));
        }
        MG_RCCL(ncclGroupEnd());
    }
    for (int g = 0; g < P; ++g)            // partial gain + own inverse transforms; the first device adds the loss term
        m.ops[g]->collidePartial(g == 0 ? Q : m.Q_rep[g], g == 0 ? f_in : m.f_rep[g], g == 0, m.streams[g]);
    if (m.use_rccl) {                      // the ONE collective of an evaluation: sum of the real Q into the caller's Q
        MG_RCCL(ncclGroupStart());
        for (int g = 0; g < P; ++g) {
            double* buf = g == 0 ? Q : m.Q_rep[g];
            MG_RCCL(ncclReduce(buf, buf, G, ncclDouble, ncclSum, 0, m.comms[g], m.streams[g]));
        }
        MG_RCCL(ncclGroupEnd());
    }
    for (int g = 0; g < P; ++g) {
        MG_HIP(hipSetDevice(m.devs[g]));
        MG_HIP(hipStreamSynchronize(m.streams[g]));
    }
    MG_HIP(hipSetDevice(prev));
}
