#include "Collisions/HIPMultiGPUBoltzmannOperator.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <iostream>
#include <mutex>
#include <thread>

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

// Worker-thread variants: a failure inside a device thread prints the same message and ends the process at once
// (std::exit from a secondary thread would run static destructors under the other threads' feet).
#define MGT_HIP(call)                                                                                             \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            std::cerr << "HIP Error: " << hipGetErrorString(e_) << " at " << __FILE__ << ":" << __LINE__ << std::endl; \
            std::_Exit(EXIT_FAILURE);                                                                             \
        }                                                                                                         \
    } while (0)
#define MGT_RCCL(call)                                                                                             \
    do {                                                                                                           \
        ncclResult_t r_ = (call);                                                                                  \
        if (r_ != ncclSuccess) {                                                                                   \
            std::cerr << "RCCL Error: " << ncclGetErrorString(r_) << " at " << __FILE__ << ":" << __LINE__ << std::endl; \
            std::_Exit(EXIT_FAILURE);                                                                              \
        }                                                                                                          \
    } while (0)

struct BoltzmannOperator<HIP_MultiGPU_Backend>::Impl {
    std::shared_ptr<GaussLegendreQuadrature> gl;
    std::shared_ptr<SphericalQuadrature> sph;
    int Nvx, Nvy, Nvz;
    double gamma, b_gamma, L;
    std::vector<int> devs;                 // as requested by setDevices(); read by the NEXT initialize() only
    int precision = 64;
    bool exact = false, hermitian = false, force_collectives = false;

    bool ready = false, use_rccl = false;
    std::vector<int> active;               // the device list initialize() actually used (sizes everything below)
    std::vector<std::unique_ptr<BoltzmannOperator<HIP_Backend>>> ops;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    std::vector<double*> f_rep, Q_rep;     // replicas on devices 1..P-1 (entry 0 unused: the caller's buffers)

    // One host thread per device: every device's ~8 kernel launches and its two RCCL calls are issued concurrently
    // instead of from one thread in turn (8 x 8 serial launches would be of the order of a 1/8 shard's run time).
    // A call publishes (Q, f) and bumps `epoch`; each worker runs its device's sequence, waits for its own stream
    // and counts itself in `done`.  Workers spin briefly after a call (time steppers call back to back), then sleep.
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<unsigned long long> epoch{0};
    std::atomic<int> done{0};
    std::atomic<bool> quit{false};
    double* cur_Q = nullptr;
    const double* cur_f = nullptr;
    size_t G = 0;

    void run_device(int g) {
        double* Qg = g == 0 ? cur_Q : Q_rep[g];
        double* fg = g == 0 ? const_cast<double*>(cur_f) : f_rep[g];
        // f: first device -> all (in place on the root).  streams[0] is an ordinary (blocking) stream, so it is ordered
        // after whatever the caller enqueued on the first device's default stream to produce f.
        if (use_rccl) MGT_RCCL(ncclBroadcast(fg, fg, G, ncclDouble, 0, comms[g], streams[g]));
        // partial gain + own inverse transforms; the first device also subtracts the loss term
        ops[g]->collidePartial(Qg, fg, g == 0, streams[g]);
        // the ONE collective of an evaluation: sum of the real Q into the caller's Q
        if (use_rccl) MGT_RCCL(ncclReduce(Qg, Qg, G, ncclDouble, ncclSum, 0, comms[g], streams[g]));
        MGT_HIP(hipStreamSynchronize(streams[g]));
    }

    void worker_main(int g) {
        MGT_HIP(hipSetDevice(active[g]));
        unsigned long long seen = 0;
        for (;;) {
            const auto t0 = std::chrono::steady_clock::now();
            while (epoch.load(std::memory_order_acquire) == seen && !quit.load(std::memory_order_acquire)) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return epoch.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_acquire); });
                    break;
                }
                std::this_thread::yield();
            }
            if (quit.load(std::memory_order_acquire)) return;
            seen = epoch.load(std::memory_order_acquire);
            run_device(g);
            done.fetch_add(1, std::memory_order_release);
        }
    }

    void release() {
        if (!ready) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit.store(true, std::memory_order_release);
        }
        cv.notify_all();
        for (std::thread& t : workers) if (t.joinable()) t.join();
        workers.clear();
        quit.store(false, std::memory_order_release);
        int prev = 0;
        (void)hipGetDevice(&prev);
        for (size_t g = 0; g < ops.size(); ++g) {          // sized by initialize(), not by a later setDevices()
            (void)hipSetDevice(active[g]);
            ops[g].reset();
            if (f_rep[g]) (void)hipFree(f_rep[g]);
            if (Q_rep[g]) (void)hipFree(Q_rep[g]);
            if (streams[g]) (void)hipStreamDestroy(streams[g]);
            if (use_rccl && comms[g]) ncclCommDestroy(comms[g]);
        }
        (void)hipSetDevice(prev);
        ops.clear(); streams.clear(); comms.clear(); f_rep.clear(); Q_rep.clear(); active.clear();
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
const std::vector<int>& BoltzmannOperator<HIP_MultiGPU_Backend>::devices() const { return impl_->ready ? impl_->active : impl_->devs; }

void BoltzmannOperator<HIP_MultiGPU_Backend>::initialize() {
    Impl& m = *impl_;
    m.release();
    int ndev = 0;
    MG_HIP(hipGetDeviceCount(&ndev));
    std::vector<int> use = m.devs;
    if (use.empty())
        for (int g = 0; g < ndev; ++g) use.push_back(g);
    for (size_t g = 0; g < use.size(); ++g) {
        const bool dup = std::count(use.begin(), use.end(), use[g]) != 1;
        if (use[g] < 0 || use[g] >= ndev || dup) {
            std::cerr << "HIP backend error in initialize: device list must name distinct visible devices (" << ndev
                      << " visible)" << std::endl;
            std::exit(EXIT_FAILURE);
        }
    }
    m.active = use;
    const int P = static_cast<int>(use.size());
    const long long B = static_cast<long long>(m.gl->getNumberOfPoints()) * m.sph->getNumberOfPoints();
    m.G = static_cast<size_t>(m.Nvx) * m.Nvy * m.Nvz;
    int prev = 0;
    MG_HIP(hipGetDevice(&prev));
    m.use_rccl = P > 1 || m.force_collectives;
    m.ops.resize(P); m.streams.assign(P, nullptr); m.comms.assign(P, nullptr);
    m.f_rep.assign(P, nullptr); m.Q_rep.assign(P, nullptr);
    m.ready = true;                       // from here on release() has something to undo
    if (m.use_rccl) MG_RCCL(ncclCommInitAll(m.comms.data(), P, m.active.data()));
    for (int g = 0; g < P; ++g) {
        MG_HIP(hipSetDevice(m.active[g]));
        MG_HIP(hipStreamCreate(&m.streams[g]));
        if (g > 0) {
            MG_HIP(hipMalloc(reinterpret_cast<void**>(&m.f_rep[g]), m.G * sizeof(double)));
            MG_HIP(hipMalloc(reinterpret_cast<void**>(&m.Q_rep[g]), m.G * sizeof(double)));
        }
        m.ops[g].reset(new BoltzmannOperator<HIP_Backend>(m.gl, m.sph, m.Nvx, m.Nvy, m.Nvz, m.gamma, m.b_gamma, m.L));
        m.ops[g]->setDevice(m.active[g]);
        m.ops[g]->setPrecision(m.precision);
        m.ops[g]->setExactReductions(m.exact, m.hermitian);
        const long long base = B / P, rem = B % P;        // contiguous, balanced shards (== bfsm.shard_range)
        const long long b0 = g * base + std::min<long long>(g, rem), b1 = b0 + base + (g < rem ? 1 : 0);
        m.ops[g]->setDirectionShard(b0, b1);
        m.ops[g]->initialize();
    }
    MG_HIP(hipSetDevice(prev));
    for (int g = 0; g < P; ++g) m.workers.emplace_back([&m, g] { m.worker_main(g); });
}

void BoltzmannOperator<HIP_MultiGPU_Backend>::computeCollision(double* Q, const double* f_in) {
    Impl& m = *impl_;
    if (!m.ready) {
        std::cerr << "HIP backend error in computeCollision: initialize() has not been called" << std::endl;
        std::exit(EXIT_FAILURE);
    }
    const int P = static_cast<int>(m.active.size());
    m.cur_Q = Q;
    m.cur_f = f_in;
    m.done.store(0, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(m.mu);
        m.epoch.fetch_add(1, std::memory_order_release);
    }
    m.cv.notify_all();
    // blocking like the reference's call (cu:218): every device thread has synchronised its own stream when it reports
    while (m.done.load(std::memory_order_acquire) < P) std::this_thread::yield();
}
