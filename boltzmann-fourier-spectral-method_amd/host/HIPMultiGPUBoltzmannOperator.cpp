#include "Collisions/HIPMultiGPUBoltzmannOperator.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "Collisions/HIPBoltzmannOperator.hpp"
#include "Collisions/detail/MultiGpuCore.hpp"

// The device side of the multi-GPU operator: HIP for devices / streams / memory, RCCL for the two collectives of an
// evaluation.  The thread choreography and the shard arithmetic live in Collisions/detail/MultiGpuCore.hpp, which is
// also instantiated (with an in-process stand-in for this struct) by the CPU test of the P > 1 path.
namespace {

struct HipRcclRuntime {
    using Stream = hipStream_t;
    using Event = hipEvent_t;
    using Comm = ncclComm_t;
    using Counters = bfsm_counters;
    using Operator = BoltzmannOperator<HIP_Backend>;

    static const char* hip(hipError_t e) { return e == hipSuccess ? nullptr : hipGetErrorString(e); }
    static const char* rccl(ncclResult_t r) { return r == ncclSuccess ? nullptr : ncclGetErrorString(r); }

    static int device_count() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
    static int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
    static const char* set_device(int d) { return hip(hipSetDevice(d)); }
    static const char* alloc_doubles(double** p, size_t n) { return hip(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(double))); }
    static void free_doubles(double* p) { (void)hipFree(p); }
    static const char* stream_create(Stream* s) { return hip(hipStreamCreate(s)); }
    static void stream_destroy(Stream s) { (void)hipStreamDestroy(s); }
    static const char* stream_sync(Stream s) { return hip(hipStreamSynchronize(s)); }
    static void* stream_handle(Stream s) { return static_cast<void*>(s); }
    static const char* event_create(Event* e) { return hip(hipEventCreateWithFlags(e, hipEventDisableTiming)); }
    static void event_destroy(Event e) { if (e) (void)hipEventDestroy(e); }
    static const char* event_record_on(Event e, void* producer) { return hip(hipEventRecord(e, static_cast<hipStream_t>(producer))); }
    static const char* stream_wait(Stream s, Event e) { return hip(hipStreamWaitEvent(s, e, 0)); }
    static const char* comm_init_all(Comm* c, int P, const int* devices) { return rccl(ncclCommInitAll(c, P, devices)); }
    static void comm_destroy(Comm c) { if (c) (void)ncclCommDestroy(c); }
    static const char* broadcast(double* buf, size_t n, int root, Comm c, Stream s) {
        return rccl(ncclBroadcast(buf, buf, n, ncclDouble, root, c, s));
    }
    static const char* reduce_sum(double* buf, size_t n, int root, Comm c, Stream s) {
        return rccl(ncclReduce(buf, buf, n, ncclDouble, ncclSum, root, c, s));
    }
    static std::unique_ptr<Operator> make_operator(std::shared_ptr<GaussLegendreQuadrature> gl, std::shared_ptr<SphericalQuadrature> sph,
                                                   int nvx, int nvy, int nvz, double gamma, double b_gamma, double L) {
        return std::unique_ptr<Operator>(new Operator(std::move(gl), std::move(sph), nvx, nvy, nvz, gamma, b_gamma, L));
    }
};

}  // namespace

struct BoltzmannOperator<HIP_MultiGPU_Backend>::Impl : bfsm_host::MultiGpuCore<HipRcclRuntime> {};

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
void BoltzmannOperator<HIP_MultiGPU_Backend>::setMaxChunk(int n) { impl_->max_chunk = n; }
void BoltzmannOperator<HIP_MultiGPU_Backend>::setMaxBatch(int n) { impl_->max_batch = n; }
void BoltzmannOperator<HIP_MultiGPU_Backend>::setProfiling(bool on) { impl_->profiling = on; }
void BoltzmannOperator<HIP_MultiGPU_Backend>::setInputStream(void* s) { impl_->input_stream = s; impl_->has_input_stream = true; }
void BoltzmannOperator<HIP_MultiGPU_Backend>::clearInputStream() { impl_->input_stream = nullptr; impl_->has_input_stream = false; }
void BoltzmannOperator<HIP_MultiGPU_Backend>::setTimeoutSeconds(double s) { impl_->timeout_s = s; }
bfsm_counters BoltzmannOperator<HIP_MultiGPU_Backend>::counters(int index) const { return impl_->counters(index); }
void BoltzmannOperator<HIP_MultiGPU_Backend>::computeCollisionBatch(double* Q, const double* f_in, int n_batch) {
    impl_->compute_batch(Q, f_in, n_batch);
}
const std::vector<int>& BoltzmannOperator<HIP_MultiGPU_Backend>::devices() const { return impl_->ready ? impl_->active : impl_->devs; }

void BoltzmannOperator<HIP_MultiGPU_Backend>::initialize() { impl_->initialize(); }

void BoltzmannOperator<HIP_MultiGPU_Backend>::computeCollision(double* Q, const double* f_in) { impl_->compute(Q, f_in); }
