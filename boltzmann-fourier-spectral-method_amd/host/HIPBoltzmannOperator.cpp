#include "Collisions/HIPBoltzmannOperator.hpp"

#include <cstdlib>
#include <iostream>

BoltzmannOperator<HIP_Backend>::BoltzmannOperator(std::shared_ptr<GaussLegendreQuadrature> gl,
                                                  std::shared_ptr<SphericalQuadrature> sph,
                                                  int nvx, int nvy, int nvz, double gamma_, double b_gamma_, double L_)
    : Nvx(nvx), Nvy(nvy), Nvz(nvz), gamma(gamma_), b_gamma(b_gamma_), L(L_),
      gl_quadrature(std::move(gl)), spherical_quadrature(std::move(sph)) {}

BoltzmannOperator<HIP_Backend>::~BoltzmannOperator() {
    if (handle_) bfsm_destroy(handle_);
}

// Same observable failure mode as HANDLE_CUDA_ERROR / CUFFT_CALL (reference CUDABoltzmannOperator.hpp:20-38).
void BoltzmannOperator<HIP_Backend>::check(int rc, const char* what) const {
    if (rc == BFSM_OK) return;
    std::cerr << "HIP backend error in " << what << " (code " << rc << "): " << bfsm_last_error(handle_) << std::endl;
    std::exit(EXIT_FAILURE);
}

void BoltzmannOperator<HIP_Backend>::initialize() {
    bfsm_desc d{};
    d.nvx = Nvx; d.nvy = Nvy; d.nvz = Nvz;
    d.n_gl = gl_quadrature->getNumberOfPoints();
    d.n_sph = spherical_quadrature->getNumberOfPoints();
    d.gl_nodes = gl_quadrature->getNodes().data();
    d.gl_wts = gl_quadrature->getWeights().data();
    d.sph_wts = spherical_quadrature->getWeights().data();
    d.sx = spherical_quadrature->getx().data();
    d.sy = spherical_quadrature->gety().data();
    d.sz = spherical_quadrature->getz().data();
    d.gamma = gamma; d.b_gamma = b_gamma; d.L = L;
    d.precision = precision_; d.device = device_;
    d.dir_begin = dir_begin_; d.dir_end = dir_end_;
    d.max_chunk = max_chunk_; d.flags = flags_; d.max_batch = max_batch_;
    if (handle_) { bfsm_destroy(handle_); handle_ = nullptr; }
    check(bfsm_create(&d, &handle_), "initialize");
}

void BoltzmannOperator<HIP_Backend>::computeCollision(double* Q, const double* f_in) {
    check(bfsm_collide(handle_, Q, f_in), "computeCollision");
}

void BoltzmannOperator<HIP_Backend>::computeCollisionBatch(double* Q, const double* f_in, int n_batch) {
    check(bfsm_collide_batch(handle_, Q, f_in, n_batch), "computeCollisionBatch");
}

void BoltzmannOperator<HIP_Backend>::collideBatchPartial(double* Q, const double* f_in, int n_batch, bool with_loss, void* stream) {
    check(bfsm_collide_batch_partial_async(handle_, Q, f_in, n_batch, with_loss ? 1 : 0, stream), "collideBatchPartial");
}

void BoltzmannOperator<HIP_Backend>::gainPartial(const double* f_in, void* stream) {
    check(bfsm_gain_partial(handle_, f_in, stream), "gainPartial");
}

void BoltzmannOperator<HIP_Backend>::finish(double* Q, const double* f_in, void* stream) {
    check(bfsm_finish(handle_, Q, f_in, stream), "finish");
}

void BoltzmannOperator<HIP_Backend>::finishPartial(double* Q, const double* f_in, bool with_loss, void* stream) {
    check(bfsm_finish_partial(handle_, Q, f_in, with_loss ? 1 : 0, stream), "finishPartial");
}

void BoltzmannOperator<HIP_Backend>::collidePartial(double* Q, const double* f_in, bool with_loss, void* stream) {
    check(bfsm_collide_partial_async(handle_, Q, f_in, with_loss ? 1 : 0, stream), "collidePartial");
}

int BoltzmannOperator<HIP_Backend>::collidePartialStatus(double* Q, const double* f_in, bool with_loss, void* stream) noexcept {
    return bfsm_collide_partial_async(handle_, Q, f_in, with_loss ? 1 : 0, stream);
}

int BoltzmannOperator<HIP_Backend>::collideBatchPartialStatus(double* Q, const double* f_in, int n_batch, bool with_loss, void* stream) noexcept {
    return bfsm_collide_batch_partial_async(handle_, Q, f_in, n_batch, with_loss ? 1 : 0, stream);
}

const char* BoltzmannOperator<HIP_Backend>::lastError() const noexcept { return bfsm_last_error(handle_); }

void* BoltzmannOperator<HIP_Backend>::qhatBuffer(size_t* n_elems, int* precision) const {
    return bfsm_qhat_buffer(handle_, n_elems, precision);
}

void BoltzmannOperator<HIP_Backend>::synchronize() { check(bfsm_synchronize(handle_), "synchronize"); }

bfsm_counters BoltzmannOperator<HIP_Backend>::counters() const {
    bfsm_counters c{};
    check(bfsm_get_counters(handle_, &c), "counters");
    return c;
}
