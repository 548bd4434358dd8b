// MultiGpuCore<Runtime> -- the device-independent part of BoltzmannOperator<HIP_MultiGPU_Backend>: direction shards,
// one host thread per device, the per-evaluation sequence (broadcast f -> partial evaluation -> ONE reduce of Q) and
// the hand-off between the calling thread and the device threads.  Everything that touches a device goes through the
// `Runtime` policy, so the very same choreography is compiled twice:
//   * host/HIPMultiGPUBoltzmannOperator.cpp instantiates it with HIP + RCCL (the product);
//   * tests/host/test_multigpu_choreography.cpp instantiates it with an in-process stand-in for the devices and the
//     two collectives (test only), which is how the P > 1 thread choreography is exercised -- and run under
//     ThreadSanitizer -- without a multi-GPU node.
//
// Runtime provides (all static; a `const char*` result is nullptr on success, else the error text):
//   types      Stream, Event, Comm, Counters, Operator   (Operator: the single-device operator; setDevice / setPrecision /
//              setExactReductions / setDirectionShard / setMaxChunk / setMaxBatch / setProfiling / initialize /
//              int collideBatchPartialStatus(Q, f, n_batch, with_loss, Stream) / const char* lastError() /
//              Counters counters())
//   devices    int device_count(); int current_device(); const char* set_device(int)
//   memory     const char* alloc_doubles(double**, size_t); void free_doubles(double*)
//   streams    const char* stream_create(Stream*); void stream_destroy(Stream); const char* stream_sync(Stream);
//              static void* stream_handle(Stream)
//   events     const char* event_create(Event*); void event_destroy(Event);
//              const char* event_record_on(Event, void* producer_stream_handle); const char* stream_wait(Stream, Event)
//   collective const char* comm_init_all(Comm*, int P, const int* devices); void comm_destroy(Comm);
//              const char* broadcast(double* buf, size_t n, int root, Comm, Stream);
//              const char* reduce_sum(double* buf, size_t n, int root, Comm, Stream)     (in place, into root's buf)
//   make_operator(gl, sph, Nvx, Nvy, Nvz, gamma, b_gamma, L) -> std::unique_ptr<Operator>
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../Quadratures/GaussLegendre.hpp"
#include "../../Quadratures/SphericalDesign.hpp"

namespace bfsm_host {

template <class RT>
class MultiGpuCore {
public:
    std::shared_ptr<GaussLegendreQuadrature> gl;
    std::shared_ptr<SphericalQuadrature> sph;
    int Nvx = 0, Nvy = 0, Nvz = 0;
    double gamma = 0, b_gamma = 0, L = 0;
    std::vector<int> devs;                 // as requested by setDevices(); read by the NEXT initialize() only
    int precision = 64;
    bool exact = false, hermitian = false, force_collectives = false;
    int max_chunk = 0;                     // directions resident at once per device (0: the library's default)
    int max_batch = 1;                     // distributions per call (compute_batch); replicas are sized for it
    bool profiling = false;                // per-kernel events on every device's operator (counters(g))
    // The stream of the FIRST device on which the caller produces f, when that is not the legacy default stream: the
    // broadcast is ordered behind an event recorded on it at the call.  Without it the first device's (blocking) stream
    // is ordered behind the legacy default stream only -- f produced on a non-blocking stream must be complete, or this
    // must be set.
    void* input_stream = nullptr;
    bool has_input_stream = false;
    // Watchdog of a blocking call: a device thread that has not reported after this many seconds (a collective that one
    // rank never joined, a hung device) ends the process with a message instead of blocking the caller for ever.  0: off.
    double timeout_s = 300.0;

    bool ready = false, use_coll = false;
    std::vector<int> active;               // the device list initialize() actually used (sizes everything below)

    ~MultiGpuCore() { release(); }

    // Same observable failure mode as the reference's HANDLE_CUDA_ERROR (CUDABoltzmannOperator.hpp:20-38): message,
    // exit.  From a device thread the process ends at once (std::exit from a secondary thread would run static
    // destructors and the runtime's teardown under the other device threads' feet).
    [[noreturn]] static void fatal(const char* what, const char* msg, bool from_worker) {
        std::cerr << "HIP backend error in " << what << ": " << (msg ? msg : "?") << std::endl;
        if (from_worker) std::_Exit(EXIT_FAILURE);
        std::exit(EXIT_FAILURE);
    }
    static void must(const char* err, const char* what, bool from_worker = false) {
        if (err) fatal(what, err, from_worker);
    }

    void initialize() {
        release();
        const int ndev = RT::device_count();
        std::vector<int> use = devs;
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
        if (use.empty()) fatal("initialize", "no device", false);
        active = use;
        const int P = static_cast<int>(use.size());
        const long long B = static_cast<long long>(gl->getNumberOfPoints()) * sph->getNumberOfPoints();
        G = static_cast<size_t>(Nvx) * Nvy * Nvz;
        const int prev = RT::current_device();
        use_coll = P > 1 || force_collectives;
        ops.resize(P); streams.assign(P, typename RT::Stream{}); comms.assign(P, typename RT::Comm{});
        have_stream.assign(P, 0); have_comm = false; have_event = false;
        f_rep.assign(P, nullptr); Q_rep.assign(P, nullptr);
        nb_cap = max_batch > 1 ? max_batch : 1;
        ready = true;                      // from here on release() has something to undo
        if (use_coll) { must(RT::comm_init_all(comms.data(), P, active.data()), "initialize (communicator)"); have_comm = true; }
        for (int g = 0; g < P; ++g) {
            must(RT::set_device(active[g]), "initialize (set device)");
            must(RT::stream_create(&streams[g]), "initialize (stream)");
            have_stream[g] = 1;
            if (g == 0) { must(RT::event_create(&input_event), "initialize (event)"); have_event = true; }
            if (g > 0) {
                must(RT::alloc_doubles(&f_rep[g], (size_t)nb_cap * G), "initialize (replica of f)");
                must(RT::alloc_doubles(&Q_rep[g], (size_t)nb_cap * G), "initialize (replica of Q)");
            }
            ops[g] = RT::make_operator(gl, sph, Nvx, Nvy, Nvz, gamma, b_gamma, L);
            ops[g]->setDevice(active[g]);
            ops[g]->setPrecision(precision);
            ops[g]->setExactReductions(exact, hermitian);
            ops[g]->setMaxChunk(max_chunk);
            ops[g]->setMaxBatch(nb_cap);
            ops[g]->setProfiling(profiling);
            const long long base = B / P, rem = B % P;        // contiguous, balanced shards (== bfsm.shard_range)
            const long long b0 = g * base + std::min<long long>(g, rem), b1 = b0 + base + (g < rem ? 1 : 0);
            ops[g]->setDirectionShard(b0, b1);
            ops[g]->initialize();
        }
        must(RT::set_device(prev), "initialize (restore device)");
        epoch.store(0, std::memory_order_relaxed);
        done.store(0, std::memory_order_relaxed);
        for (int g = 0; g < P; ++g) workers.emplace_back([this, g] { worker_main(g); });
    }

    // Blocking like the reference's call (CUDABoltzmannOperator.cu:218): every device thread has synchronised its own
    // stream when it reports.  The caller spins briefly (a 1/8 shard of the small configurations takes a few hundred
    // microseconds), then sleeps on the condition variable instead of holding a core for tens of milliseconds.
    void compute(double* Q, const double* f_in) { compute_batch(Q, f_in, 1); }

    // n_batch distributions [n_batch][G] at once (n_batch <= max_batch at initialize()): one broadcast, one batched shard
    // evaluation per device, ONE reduce of n_batch * G reals.
    void compute_batch(double* Q, const double* f_in, int n_batch) {
        if (!ready) {
            std::cerr << "HIP backend error in computeCollision: initialize() has not been called" << std::endl;
            std::exit(EXIT_FAILURE);
        }
        if (n_batch < 1 || n_batch > nb_cap) {
            std::cerr << "HIP backend error in computeCollision: n_batch must be in [1, max_batch at initialize()]" << std::endl;
            std::exit(EXIT_FAILURE);
        }
        const int P = static_cast<int>(active.size());
        if (has_input_stream) {            // order the first device's stream behind the caller's producer of f
            const int prev = RT::current_device();
            must(RT::set_device(active[0]), "computeCollision (set device)");
            must(RT::event_record_on(input_event, input_stream), "computeCollision (event on the caller's stream)");
            must(RT::set_device(prev), "computeCollision (restore device)");
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            cur_Q = Q;
            cur_f = f_in;
            cur_nb = n_batch;
            cur_wait = has_input_stream;
            done.store(0, std::memory_order_relaxed);
            epoch.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
        const auto t0 = std::chrono::steady_clock::now();
        const bool watch = timeout_s > 0;
        const auto deadline = t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(watch ? timeout_s : 0.0));
        while (done.load(std::memory_order_acquire) < P) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (done.load(std::memory_order_acquire) >= P) break;
                    if (!watch) { cv_done.wait(lk, [&] { return done.load(std::memory_order_acquire) >= P; }); break; }
                    // (a system_clock deadline: the steady_clock overload waits through pthread_cond_clockwait, which the
                    // ThreadSanitizer of this toolchain does not intercept -- it then reports a double lock of `mu`)
                    const auto left = deadline - std::chrono::steady_clock::now();
                    const auto wall = std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(left);
                    if (cv_done.wait_until(lk, wall, [&] { return done.load(std::memory_order_acquire) >= P; })) break;
                    if (std::chrono::steady_clock::now() < deadline) continue;        // the wall clock jumped: keep waiting
                    // the device threads that have not reported are parked inside a collective or a stream wait: nothing
                    // can be unwound from here, so the process ends with the message (same policy as a device-side error)
                    const int n_done = done.load(std::memory_order_acquire);
                    const std::string msg = std::to_string(P - n_done) + " of " + std::to_string(P) +
                                            " device thread(s) did not finish within " + std::to_string(timeout_s) +
                                            " s (a collective that one rank never joined, or a hung device)";
                    fatal("computeCollision (watchdog)", msg.c_str(), true);
                }
                break;
            }
            std::this_thread::yield();
        }
    }

    // Counters of the device-th operator of the active team (per-kernel times need profiling = true at initialize()).
    typename RT::Counters counters(int index) const {
        if (!ready || index < 0 || index >= static_cast<int>(ops.size())) {
            std::cerr << "HIP backend error in counters: no such device in the active team" << std::endl;
            std::exit(EXIT_FAILURE);
        }
        return ops[index]->counters();
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
        const int prev = RT::current_device();
        for (size_t g = 0; g < ops.size(); ++g) {          // sized by initialize(), not by a later setDevices()
            (void)RT::set_device(active[g]);
            ops[g].reset();
            if (f_rep[g]) RT::free_doubles(f_rep[g]);
            if (Q_rep[g]) RT::free_doubles(Q_rep[g]);
            if (have_stream[g]) RT::stream_destroy(streams[g]);
            if (have_comm) RT::comm_destroy(comms[g]);
            if (g == 0 && have_event) { RT::event_destroy(input_event); have_event = false; }
        }
        (void)RT::set_device(prev);
        ops.clear(); streams.clear(); comms.clear(); f_rep.clear(); Q_rep.clear(); active.clear(); have_stream.clear();
        have_comm = false;
        ready = false;
    }

private:
    std::vector<std::unique_ptr<typename RT::Operator>> ops;
    std::vector<typename RT::Stream> streams;
    std::vector<typename RT::Comm> comms;
    std::vector<char> have_stream;
    bool have_comm = false, have_event = false;
    typename RT::Event input_event{};      // recorded on the caller's stream (input_stream) at every call, first device
    int nb_cap = 1;
    std::vector<double*> f_rep, Q_rep;     // replicas on devices 1..P-1 (entry 0 unused: the caller's buffers)

    // One host thread per device: every device's ~8 kernel launches and its two collective calls are issued
    // concurrently instead of from one thread in turn (8 x 8 serial launches would be of the order of a 1/8 shard's
    // run time).  A call publishes (Q, f) and bumps `epoch` under the mutex; each worker runs its device's sequence,
    // waits for its own stream and counts itself in `done`; the last one wakes the caller.  Workers spin briefly after
    // a call (time steppers call back to back), then sleep.
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::atomic<unsigned long long> epoch{0};
    std::atomic<int> done{0};
    std::atomic<bool> quit{false};
    double* cur_Q = nullptr;               // written under `mu` before the epoch bump, read by the workers after they
    const double* cur_f = nullptr;         // have observed the new epoch (acquire)
    int cur_nb = 1;
    bool cur_wait = false;
    size_t G = 0;

    void run_device(int g) {
        double* Qg = g == 0 ? cur_Q : Q_rep[g];
        double* fg = g == 0 ? const_cast<double*>(cur_f) : f_rep[g];
        const size_t n = static_cast<size_t>(cur_nb) * G;
        // f: first device -> all (in place on the root).  streams[0] is an ordinary (blocking) stream, so it is ordered
        // after whatever the caller enqueued on the first device's legacy default stream to produce f; a producer on any
        // other stream is named by input_stream and waited for through the event recorded at the call.
        if (g == 0 && cur_wait) must(RT::stream_wait(streams[0], input_event), "computeCollision (wait for the caller's stream)", true);
        if (use_coll) must(RT::broadcast(fg, n, 0, comms[g], streams[g]), "computeCollision (broadcast of f)", true);
        // partial gain + own inverse transforms; the first device also subtracts the loss term.  The status variant
        // of the operator call: a failure ends the process from here (message first), never through std::exit on a
        // device thread.
        if (ops[g]->collideBatchPartialStatus(Qg, fg, cur_nb, g == 0, RT::stream_handle(streams[g])) != 0)
            fatal("computeCollision (device shard)", ops[g]->lastError(), true);
        // the ONE collective of an evaluation: sum of the real Q into the caller's Q
        if (use_coll) must(RT::reduce_sum(Qg, n, 0, comms[g], streams[g]), "computeCollision (reduce of Q)", true);
        must(RT::stream_sync(streams[g]), "computeCollision (stream synchronize)", true);
    }

    void worker_main(int g) {
        must(RT::set_device(active[g]), "device thread (set device)", true);
        const int P = static_cast<int>(active.size());
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
            if (done.fetch_add(1, std::memory_order_acq_rel) + 1 == P) {
                std::lock_guard<std::mutex> lk(mu);      // pairs with the caller's wait: no lost wake-up
                cv_done.notify_all();
            }
        }
    }
};

}  // namespace bfsm_host
