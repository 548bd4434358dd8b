// TEST ONLY.  The P > 1 choreography of BoltzmannOperator<HIP_MultiGPU_Backend> on the CPU.
//
// host/Collisions/detail/MultiGpuCore.hpp holds everything of the multi-GPU operator that is not a HIP / RCCL call:
// shards, one host thread per device, the hand-off between the caller and the device threads, the per-evaluation
// sequence broadcast(f) -> partial evaluation -> ONE reduce(Q).  Here it is instantiated with an in-process stand-in
// for the devices (plain host memory, synchronous "streams"), for the two collectives (rendezvous of the P device
// threads on a barrier: like RCCL, a collective completes only when every rank has called it) and for the per-device
// operator (an exactly representable function of (f, shard), so the sum over the shards must equal the single-device
// result bit for bit).  Built twice by tests/test_host_mirror.py: plain, and with -fsanitize=thread.
// Failure modes (second argument): "fail-broadcast" / "fail-reduce" -- the collective returns an error on ONE rank;
// "fail-shard" -- one device's operator reports a failure; "stall" -- one rank never joins the reduce.  The process must end
// with the message and a non-zero status (never hang in compute()): the test runs each mode as a child process.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "Collisions/detail/MultiGpuCore.hpp"

static int failures = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

namespace fake {

constexpr int N_DEVICES = 8;
thread_local int tl_device = 0;
std::atomic<int> live_streams{0}, live_buffers{0}, live_comms{0}, live_ops{0}, wrong_device{0}, live_events{0};
std::atomic<int> events_recorded{0}, waits_before_record{0}, chunk_seen{-1}, batch_seen{-1}, profiling_seen{-1};
// failure injection (child-process modes): which call fails / stalls, and on which rank
enum Fault { NONE, FAIL_BROADCAST, FAIL_REDUCE, FAIL_SHARD, STALL_REDUCE };
Fault fault = NONE;
int fault_rank = 1;

struct Team {                       // what ncclCommInitAll creates: P ranks that rendezvous
    int P;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long long generation = 0;
    std::vector<double*> slot;
    explicit Team(int p) : P(p), slot(p, nullptr) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long gen = generation;
        if (++waiting == P) { waiting = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
struct Comm { std::shared_ptr<Team> team; int rank = -1; };

// weight of direction b at grid point i, and the loss factor: small integers, so every partial sum is exact in double
inline double wgt(long long b, size_t i) { return (double)((b * 31 + (long long)i * 7) % 13); }

struct Operator {
    std::shared_ptr<GaussLegendreQuadrature> gl;
    std::shared_ptr<SphericalQuadrature> sph;
    size_t G;
    int device = -1, precision = 0;
    bool exact = false, herm = false, initialised = false;
    long long b0 = 0, b1 = 0;
    Operator(std::shared_ptr<GaussLegendreQuadrature> g, std::shared_ptr<SphericalQuadrature> s, size_t G_) : gl(g), sph(s), G(G_) { ++live_ops; }
    ~Operator() { --live_ops; }
    void setDevice(int d) { device = d; }
    void setPrecision(int p) { precision = p; }
    void setExactReductions(bool e, bool h) { exact = e; herm = h; }
    void setDirectionShard(long long a, long long b) { b0 = a; b1 = b; }
    int max_chunk = -1, max_batch = -1, calls = 0;
    bool profiling = false;
    void setMaxChunk(int n) { max_chunk = n; chunk_seen = n; }
    void setMaxBatch(int n) { max_batch = n; batch_seen = n; }
    void setProfiling(bool on) { profiling = on; profiling_seen = on ? 1 : 0; }
    void initialize() { if (tl_device != device) ++wrong_device; initialised = true; }
    int collideBatchPartialStatus(double* Q, const double* f, int nb, bool with_loss, void*) noexcept {
        if (tl_device != device || !initialised) ++wrong_device;      // issued from the thread bound to this device
        if (nb < 1 || nb > max_batch) return 2;
        if (fault == FAIL_SHARD && device == fault_rank) return 7;
        ++calls;
        for (int m = 0; m < nb; ++m)
            for (size_t i = 0; i < G; ++i) {
                double s = 0;
                for (long long b = b0; b < b1; ++b) s += wgt(b, i) * f[m * G + i];
                Q[m * G + i] = with_loss ? s - 3.0 * f[m * G + i] : s;
            }
        return 0;
    }
    const char* lastError() const noexcept { return "injected shard failure"; }
    struct Counters { long long b0, b1; int calls, device; };
    Counters counters() const { return Counters{b0, b1, calls, device}; }
};

struct Runtime {
    using Stream = int;
    struct Event { std::shared_ptr<std::atomic<int>> recorded; };
    using Comm = fake::Comm;
    using Operator = fake::Operator;
    using Counters = fake::Operator::Counters;
    static const char* event_create(Event* e) { e->recorded = std::make_shared<std::atomic<int>>(0); ++live_events; return nullptr; }
    static void event_destroy(Event) { --live_events; }
    static const char* event_record_on(Event e, void* producer) {
        if (producer != reinterpret_cast<void*>(0x5EED)) return "event recorded on the wrong stream";
        e.recorded->fetch_add(1);
        ++events_recorded;
        return nullptr;
    }
    static const char* stream_wait(Stream, Event e) {      // the record of THIS call must already have happened
        if (e.recorded->exchange(0) < 1) ++waits_before_record;
        return nullptr;
    }
    static int device_count() { return N_DEVICES; }
    static int current_device() { return tl_device; }
    static const char* set_device(int d) { if (d < 0 || d >= N_DEVICES) return "invalid device"; tl_device = d; return nullptr; }
    static const char* alloc_doubles(double** p, size_t n) { *p = new double[n]; ++live_buffers; return nullptr; }
    static void free_doubles(double* p) { delete[] p; --live_buffers; }
    static const char* stream_create(Stream* s) { *s = 1; ++live_streams; return nullptr; }
    static void stream_destroy(Stream) { --live_streams; }
    static const char* stream_sync(Stream) { return nullptr; }
    static void* stream_handle(Stream) { return nullptr; }
    static const char* comm_init_all(Comm* c, int P, const int*) {
        auto t = std::make_shared<Team>(P);
        for (int r = 0; r < P; ++r) { c[r].team = t; c[r].rank = r; ++live_comms; }
        return nullptr;
    }
    static void comm_destroy(Comm) { --live_comms; }
    static const char* broadcast(double* buf, size_t n, int root, Comm c, Stream) {
        if (fault == FAIL_BROADCAST && c.rank == fault_rank) return "injected broadcast failure";
        Team& t = *c.team;
        t.slot[c.rank] = buf;
        t.barrier();
        if (c.rank != root) std::memcpy(buf, t.slot[root], n * sizeof(double));
        t.barrier();
        return nullptr;
    }
    static const char* reduce_sum(double* buf, size_t n, int root, Comm c, Stream) {
        if (fault == FAIL_REDUCE && c.rank == fault_rank) return "injected reduce failure";
        if (fault == STALL_REDUCE && c.rank == fault_rank)          // this rank never joins: the others wait in the rendezvous
            for (;;) std::this_thread::sleep_for(std::chrono::seconds(3600));
        Team& t = *c.team;
        t.slot[c.rank] = buf;
        t.barrier();
        if (c.rank == root)
            for (int r = 0; r < t.P; ++r)
                if (r != root) for (size_t i = 0; i < n; ++i) buf[i] += t.slot[r][i];
        t.barrier();
        return nullptr;
    }
    static std::unique_ptr<Operator> make_operator(std::shared_ptr<GaussLegendreQuadrature> gl, std::shared_ptr<SphericalQuadrature> sph,
                                                   int nx, int ny, int nz, double, double, double) {
        return std::unique_ptr<Operator>(new Operator(gl, sph, (size_t)nx * ny * nz));
    }
};

}  // namespace fake

using Core = bfsm_host::MultiGpuCore<fake::Runtime>;

static void expected(std::vector<double>& Q, const std::vector<double>& f, long long B) {
    for (size_t i = 0; i < f.size(); ++i) {
        double s = 0;
        for (long long b = 0; b < B; ++b) s += fake::wgt(b, i) * f[i];
        Q[i] = s - 3.0 * f[i];
    }
}

static void fill(std::vector<double>& f, int k) {
    for (size_t i = 0; i < f.size(); ++i) f[i] = (double)((i * 5 + (size_t)k * 11) % 17) - 8.0;
}

static void setup(Core& c, const std::string& design_dir, int n_gl, int n_sph, int nv) {
    c.gl = std::make_shared<GaussLegendreQuadrature>(n_gl, 0.0, 1.0);
    c.sph = std::make_shared<SphericalDesign>(n_sph, design_dir);
    c.Nvx = c.Nvy = c.Nvz = nv;
    c.gamma = 0; c.b_gamma = 1; c.L = 1;
}

static void run_calls(Core& c, int calls, long long B, size_t G, int salt) {
    std::vector<double> f(G), Q(G), ref(G);
    for (int k = 0; k < calls; ++k) {                       // back to back, f changes every call
        fill(f, k + salt);
        std::fill(Q.begin(), Q.end(), -1.0);
        c.compute(Q.data(), f.data());
        expected(ref, f, B);
        CHECK(std::memcmp(Q.data(), ref.data(), G * sizeof(double)) == 0);
    }
}

static void run_batches(Core& c, int nb, long long B, size_t G, int salt) {
    std::vector<double> f(G * nb), Q(G * nb), ref(G), one(G);
    for (int m = 0; m < nb; ++m) { fill(one, salt + 3 * m); std::copy(one.begin(), one.end(), f.begin() + m * G); }
    std::fill(Q.begin(), Q.end(), -1.0);
    c.compute_batch(Q.data(), f.data(), nb);
    for (int m = 0; m < nb; ++m) {
        fill(one, salt + 3 * m);
        expected(ref, one, B);
        CHECK(std::memcmp(Q.data() + m * G, ref.data(), G * sizeof(double)) == 0);
    }
}

int main(int argc, char** argv) {
    const std::string design_dir = argc > 1 ? argv[1] : "";
    const std::string mode = argc > 2 ? argv[2] : "";
    const int n_gl = 5, n_sph = 12, nv = 6;                 // B = 60: not a multiple of 8 (uneven shards)
    const long long B = (long long)n_gl * n_sph;
    const size_t G = (size_t)nv * nv * nv;

    if (!mode.empty()) {        // failure modes: must end the process with a message and a non-zero status, never hang
        fake::fault = mode == "fail-broadcast" ? fake::FAIL_BROADCAST : mode == "fail-reduce" ? fake::FAIL_REDUCE
                    : mode == "fail-shard" ? fake::FAIL_SHARD : mode == "stall" ? fake::STALL_REDUCE : fake::NONE;
        if (fake::fault == fake::NONE) { std::printf("unknown mode\n"); return 2; }
        Core c;
        setup(c, design_dir, n_gl, n_sph, nv);
        c.devs = {0, 1, 2, 3};
        c.timeout_s = 1.0;                                  // the watchdog that turns the stalled rank into an exit
        c.initialize();
        std::vector<double> f(G, 1.0), Q(G);
        c.compute(Q.data(), f.data());
        std::printf("compute() returned although a fault was injected\n");
        return 0;                                           // reaching this line is the failure the test looks for
    }

    for (int P : {1, 2, 3, 8}) {
        Core c;
        setup(c, design_dir, n_gl, n_sph, nv);
        c.devs.clear();
        for (int g = 0; g < P; ++g) c.devs.push_back(g);
        c.initialize();
        CHECK((int)c.active.size() == P && c.use_coll == (P > 1));
        run_calls(c, 100, B, G, P);
        c.release();
        CHECK(fake::live_streams == 0 && fake::live_buffers == 0 && fake::live_comms == 0 && fake::live_ops == 0);
    }
    {   // a device list that is neither sorted nor starting at 0; the caller's thread keeps its own device
        Core c;
        setup(c, design_dir, n_gl, n_sph, nv);
        c.devs = {5, 2, 7};
        fake::tl_device = 4;
        c.initialize();
        CHECK(fake::tl_device == 4);
        run_calls(c, 20, B, G, 77);
        // initialize() twice: the first team is torn down (threads joined, resources freed) and a new one built
        c.initialize();
        CHECK(fake::live_streams == 3 && fake::live_buffers == 4 && fake::live_comms == 3 && fake::live_ops == 3);
        run_calls(c, 20, B, G, 78);
        // setDevices() after initialize() takes effect at the NEXT initialize(); until then the active team serves
        c.devs = {0, 1, 2, 3, 4, 5, 6, 7};
        run_calls(c, 20, B, G, 79);
        CHECK(c.active.size() == 3);
        c.initialize();
        CHECK(c.active.size() == 8 && fake::live_streams == 8 && fake::live_buffers == 14 && fake::live_comms == 8);
        run_calls(c, 20, B, G, 80);
        // single device with the collectives forced on (what the one-GPU box exercises on hardware)
        c.devs = {6};
        c.force_collectives = true;
        c.initialize();
        CHECK(c.use_coll && c.active.size() == 1);
        run_calls(c, 20, B, G, 81);
    }   // destructor releases
    CHECK(fake::live_streams == 0 && fake::live_buffers == 0 && fake::live_comms == 0 && fake::live_ops == 0);
    CHECK(fake::wrong_device == 0);
    {   // knobs reach every device's operator; batches; counters; the caller's stream is waited for on the first device
        Core c;
        setup(c, design_dir, n_gl, n_sph, nv);
        c.devs = {3, 1, 6};
        c.max_chunk = 17; c.max_batch = 4; c.profiling = true;
        c.initialize();
        CHECK(fake::chunk_seen == 17 && fake::batch_seen == 4 && fake::profiling_seen == 1 && fake::live_events == 1);
        for (int nb : {1, 2, 4}) run_batches(c, nb, B, G, 40 + nb);
        run_calls(c, 5, B, G, 90);                           // single evaluations on a batch-capable team
        long long covered = 0;
        for (int g = 0; g < 3; ++g) {
            const auto cn = c.counters(g);
            CHECK(cn.device == c.active[g] && cn.calls == 8 && cn.b1 > cn.b0);
            covered += cn.b1 - cn.b0;
        }
        CHECK(covered == B);
        c.input_stream = reinterpret_cast<void*>(0x5EED); c.has_input_stream = true;
        const int before = fake::events_recorded;
        run_calls(c, 10, B, G, 91);
        CHECK(fake::events_recorded == before + 10 && fake::waits_before_record == 0);
        c.has_input_stream = false;
        run_calls(c, 3, B, G, 92);
        CHECK(fake::events_recorded == before + 10);
    }
    CHECK(fake::live_events == 0);
    {   // a sleeping team (workers parked on the condition variable) wakes up for the next call and for release()
        Core c;
        setup(c, design_dir, n_gl, n_sph, nv);
        c.devs = {0, 1, 2, 3};
        c.initialize();
        run_calls(c, 3, B, G, 5);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        run_calls(c, 3, B, G, 6);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    if (failures == 0) std::printf("multi-GPU choreography checks passed\n");
    return failures == 0 ? 0 : 1;
}
