// TEST ONLY.  The P > 1 choreography of BoltzmannOperator<HIP_MultiGPU_Backend> on the CPU.
//
// host/Collisions/detail/MultiGpuCore.hpp holds everything of the multi-GPU operator that is not a HIP / RCCL call:
// shards, one host thread per device, the hand-off between the caller and the device threads, the per-evaluation
// sequence broadcast(f) -> partial evaluation -> ONE reduce(Q).  Here it is instantiated with an in-process stand-in
// for the devices (plain host memory, synchronous "streams"), for the two collectives (rendezvous of the P device
// threads on a barrier: like RCCL, a collective completes only when every rank has called it) and for the per-device
// operator (an exactly representable function of (f, shard), so the sum over the shards must equal the single-device
// result bit for bit).  Built twice by tests/test_host_mirror.py: plain, and with -fsanitize=thread.
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
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
std::atomic<int> live_streams{0}, live_buffers{0}, live_comms{0}, live_ops{0}, wrong_device{0};

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
    void initialize() { if (tl_device != device) ++wrong_device; initialised = true; }
    int collidePartialStatus(double* Q, const double* f, bool with_loss, void*) noexcept {
        if (tl_device != device || !initialised) ++wrong_device;      // issued from the thread bound to this device
        for (size_t i = 0; i < G; ++i) {
            double s = 0;
            for (long long b = b0; b < b1; ++b) s += wgt(b, i) * f[i];
            Q[i] = with_loss ? s - 3.0 * f[i] : s;
        }
        return 0;
    }
    const char* lastError() const noexcept { return ""; }
};

struct Runtime {
    using Stream = int;
    using Comm = fake::Comm;
    using Operator = fake::Operator;
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
        Team& t = *c.team;
        t.slot[c.rank] = buf;
        t.barrier();
        if (c.rank != root) std::memcpy(buf, t.slot[root], n * sizeof(double));
        t.barrier();
        return nullptr;
    }
    static const char* reduce_sum(double* buf, size_t n, int root, Comm c, Stream) {
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

int main(int argc, char** argv) {
    const std::string design_dir = argc > 1 ? argv[1] : "";
    const int n_gl = 5, n_sph = 12, nv = 6;                 // B = 60: not a multiple of 8 (uneven shards)
    const long long B = (long long)n_gl * n_sph;
    const size_t G = (size_t)nv * nv * nv;

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
