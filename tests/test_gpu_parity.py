"""-m gpu: parity of the HIP path (through the C-ABI) against the oracle, the golden BKW norms, and
size-independent properties at BASELINE.json's full sizes.  Tolerances (fp64): max|Q - Q_ref| <= 1e-12 max|Q_ref|
(BASELINE.md section 3), |L2err - L2err_ref| <= 1e-10 (north star).  fp32 variant: 5e-6 relative (measured: 1.1e-6 on
the full config 5 quadrature)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "bkw_norms.json")))
TOL64 = 1e-12
TOL32 = 5e-6


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the HIP path has no fallback)")
    return torch


def _make(bfsm, nv, n_gl, n_sph, precision=64, shard=None, max_chunk=0, profile=False, exact=False, gamma=None,
          b_gamma=None, max_batch=0, hermitian=False, small_path=True):
    c = dict(bfsm.reference_constants())
    if gamma is not None:
        c["gamma"], c["b_gamma"] = gamma, b_gamma
    L = c["L"]
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph),
                                   nv, nv, nv, c["gamma"], c["b_gamma"], L)
    op.setPrecision(precision)
    if shard:
        op.setDirectionShard(*shard)
    if max_chunk:
        op.setMaxChunk(max_chunk)
    op.setProfiling(profile)
    op.setExactReductions(exact or hermitian, hermitian=hermitian)
    op.setMaxBatch(max_batch)
    op.setSmallPath(small_path)
    op.initialize()
    return op


def _collide(torch, op, f_h):
    f = torch.from_numpy(np.ascontiguousarray(f_h)).cuda()
    Q = torch.empty_like(f)
    torch.cuda.synchronize()
    op(Q, f)
    return Q.cpu().numpy()


def _oracle(oracle, f_h, n_gl, n_sph, **kw):
    import bfsm
    c = bfsm.reference_constants()
    return oracle.collide(f_h, oracle.gauss_legendre(n_gl, 0.0, c["R"]), oracle.spherical_design(n_sph),
                          c["gamma"], c["b_gamma"], c["L"], **kw)


def test_library_is_the_hip_one(torch_cuda):
    import bfsm
    assert bfsm.load_library().bfsm_backend_name() == b"HIP"
    assert os.path.exists(bfsm.lib_path())


@pytest.mark.parametrize("n,prec,tol", [(16, 64, 3e-15), (32, 64, 3e-15), (64, 64, 4e-15), (16, 32, 1e-6),
                                        (32, 32, 1e-6), (64, 32, 1e-6), (128, 32, 1e-6), (128, 64, 3e-15)])
def test_fft3d_kernels(torch_cuda, n, prec, tol):
    """Hand-written Stockham passes vs numpy.fft + round trip (reference check: cufft_benchmark.cu:166-207)."""
    import bfsm
    torch = torch_cuda
    op = _make(bfsm, n, 1, 6, prec)
    rng = np.random.default_rng(n)
    batch = 3
    a = rng.standard_normal((batch, n, n, n)) + 1j * rng.standard_normal((batch, n, n, n))
    ref = np.fft.fftn(a, axes=(1, 2, 3))
    cdtype = torch.complex128 if prec == 64 else torch.complex64
    d = torch.from_numpy(a).to(cdtype).cuda()
    op.fft3d(d, batch, -1)
    fw = d.cpu().numpy().astype(np.complex128).transpose(0, 1, 3, 2)          # [lx][lz][ly] -> natural
    assert np.abs(fw - ref).max() <= tol * np.abs(ref).max()
    op.fft3d(d, batch, +1)                                                     # spectral layout in, natural out
    back = d.cpu().numpy().astype(np.complex128) / n ** 3
    assert np.abs(back - a).max() <= 2 * tol * np.abs(a).max()
    op.destroy()


@pytest.mark.parametrize("nv,n_gl,n_sph,max_chunk", [(16, 8, 32, 0), (16, 3, 12, 5), (32, 8, 48, 0), (32, 3, 6, 4),
                                                     (64, 2, 12, 0)])
@pytest.mark.parametrize("inp", ["bkw", "random"])
def test_collide_matches_oracle_fp64(torch_cuda, oracle, nv, n_gl, n_sph, max_chunk, inp):
    import bfsm
    f_h, _, _, _ = bfsm.bkw_solution(nv)
    if inp == "random":
        f_h = bfsm.perturbed_input(f_h)
    op = _make(bfsm, nv, n_gl, n_sph, 64, max_chunk=max_chunk)
    got = _collide(torch_cuda, op, f_h)
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max()
    op.destroy()


def test_collide_fp32_variant(torch_cuda, oracle):
    import bfsm
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(32)[0])
    op = _make(bfsm, 32, 4, 12, 32)
    got = _collide(torch_cuda, op, f_h)
    ref = _oracle(oracle, f_h, 4, 12)
    assert np.abs(got - ref).max() <= TOL32 * np.abs(ref).max()
    op.destroy()


@pytest.mark.parametrize("row", GOLD["published"] + GOLD["survey"], ids=lambda r: f"N{r['nv']}_gl{r['n_gl']}_s{r['n_sph']}")
def test_bkw_golden_norms(torch_cuda, row):
    """The reference's own known-answer printout (Results/maxwell_bkw_fftw_atomics.txt) reproduced by the HIP path,
    including the full-size config 3 (N=64, M_gl=16, 48-point design) and the published N=64, M_gl=64 rows."""
    import bfsm
    f_h, q_exact, _, dv = bfsm.bkw_solution(row["nv"])
    op = _make(bfsm, row["nv"], row["n_gl"], row["n_sph"], 64)
    got = _collide(torch_cuda, op, f_h)
    l1, l2, linf = bfsm.error_norms(got, q_exact, dv)
    assert abs(l2 - row["L2"]) <= 1e-10                                   # north-star criterion
    for name, val in (("L1", l1), ("L2", l2), ("Linf", linf)):
        assert val == pytest.approx(row[name], rel=row.get(name + "_rtol", 6e-9)), name
    if "sum_abs_Q" in row:
        assert np.abs(got).sum() == pytest.approx(row["sum_abs_Q"], rel=2e-10)
    op.destroy()


class _DevView:
    """Zero-copy torch view of a raw device pointer (the handle-owned partial Q_gain_hat) via __cuda_array_interface__."""

    def __init__(self, ptr, n, prec):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8" if prec == 64 else "<f4",
                                         "data": (int(ptr), False), "version": 2}

    def tensor(self, torch):
        return torch.as_tensor(self, device="cuda")


def _sharded_collide(torch, bfsm, nv, n_gl, n_sph, f, P, precision=64, max_chunk=0, exact=False):
    """P handles with disjoint direction shards on ONE device; the sum of their partial Q_gain_hat buffers is what
    the RCCL all-reduce produces on a multi-GPU node; finish() on shard 0 completes the evaluation."""
    ops = [_make(bfsm, nv, n_gl, n_sph, precision, shard=bfsm.shard_range(n_gl * n_sph, r, P), max_chunk=max_chunk,
                 exact=exact) for r in range(P)]
    views = []
    for op in ops:
        op.gainPartial(f)
        op.synchronize()
        views.append(_DevView(*op.qhatBuffer()).tensor(torch))
    total = views[0].clone()
    for v in views[1:]:
        total += v
    views[0].copy_(total)
    Q = torch.empty_like(f)
    torch.cuda.synchronize()
    ops[0].finish(Q, f)
    ops[0].synchronize()
    out = Q.cpu().numpy()
    for op in ops:
        op.destroy()
    return out


def test_sharded_gain_sums_to_single_device_result(torch_cuda, oracle):
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 32, 4, 12
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    f = torch.from_numpy(f_h).cuda()
    whole = _make(bfsm, nv, n_gl, n_sph)
    a = _collide(torch, whole, f_h)
    whole.destroy()
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    for P in (2, 3, 5):
        b = _sharded_collide(torch, bfsm, nv, n_gl, n_sph, f, P, max_chunk=5)
        assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()       # differs from P=1 only by summation order
        assert np.abs(b - ref).max() <= TOL64 * np.abs(ref).max()


def test_collide_rejects_partial_handle(torch_cuda):
    import bfsm
    torch = torch_cuda
    op = _make(bfsm, 16, 2, 6, shard=(0, 5))
    f = torch.zeros(16 ** 3, dtype=torch.float64, device="cuda")
    with pytest.raises(bfsm.BfsmError):
        op(torch.empty_like(f), f)
    op.destroy()


def test_properties_at_full_size(torch_cuda):
    """Config 3 (N=64, M_gl=16, ss009.048) and config 4 (ss017.156) sizes, size-independent checks: bitwise
    determinism (no atomics), quadratic scaling Q(a f) = a^2 Q(f), and 8 direction shards == whole."""
    import bfsm
    torch = torch_cuda
    nv, n_gl = 64, 16
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    f = torch.from_numpy(f_h).cuda()
    for n_sph in (48, 156):
        op = _make(bfsm, nv, n_gl, n_sph)
        Q = torch.empty_like(f)
        op(Q, f)
        q1 = Q.cpu().numpy().copy()
        op(Q, f)
        assert np.array_equal(q1, Q.cpu().numpy())
        op(Q, f * 3.0)
        assert np.abs(Q.cpu().numpy() - 9.0 * q1).max() <= 1e-12 * np.abs(9.0 * q1).max()
        op.destroy()
        q8 = _sharded_collide(torch, bfsm, nv, n_gl, n_sph, f, 8)
        assert np.abs(q8 - q1).max() <= 1e-13 * np.abs(q1).max()


def test_fp32_config5_shape_runs_and_matches_fp64_on_a_subset(torch_cuda):
    """N=128 single-precision variant (config 5 grid): a few radial nodes of the 192-point design, compared with
    the fp64 path at N=64 being impossible, we check fp32 self-consistency: shards == whole and Q finite, and the
    BKW norms at N=128 (truncation error is negligible there, so the fp32 rounding level ~1e-6 shows directly)."""
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 128, 2, 192
    f_h, q_exact, _, dv = bfsm.bkw_solution(nv)
    f = torch.from_numpy(f_h).cuda()
    op = _make(bfsm, nv, n_gl, n_sph, 32)
    got = _collide(torch, op, f_h)
    op.destroy()
    assert np.isfinite(got).all()
    q2 = _sharded_collide(torch, bfsm, nv, n_gl, n_sph, f, 2, precision=32)
    assert np.abs(q2 - got).max() <= 1e-4 * np.abs(got).max()


def test_cpp_driver_reproduces_published_norms(torch_cuda):
    """The C++ mirror of the reference's operator class + BKW driver (host/maxwell_bkw_hip.cpp), run like the
    reference's `maxwell_bkw_cuda_ex --Nv 32 --Ns 12`, prints the norms archived in
    Results/maxwell_bkw_fftw_atomics.txt:19-21."""
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "boltzmann-fourier-spectral-method_amd", "maxwell_bkw_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s", "maxwell_bkw_hip"])
    for extra in ([], ["--hermitian"]):       # the opt-in exact reductions print the same norms
        out = subprocess.run([exe, "--Nv", "32", "--Ns", "12", "-t", "3", "--design-dir",
                              os.path.join(os.path.dirname(exe), "data", "sph_design")] + extra,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        text = out.stdout
        assert "Run statistics for HIP" in text and "Total number of samples taken: 3" in text
        got = {k: float(re.search(k + r" error: (\S+)", text).group(1)) for k in ("L1", "L2", "Linf")}
        row = GOLD["published"][0]
        for k in ("L1", "L2", "Linf"):
            # the stream is left in scientific / precision 8 like the reference's: the log line IS the archived line
            assert re.search(k + r" error: (\S+)", text).group(1) == "%.8e" % row[k], (k, got, text)


@pytest.mark.parametrize("nv,n_gl,n_sph", [(16, 8, 32), (32, 8, 48), (64, 2, 12)])
def test_exact_reductions_match_oracle(torch_cuda, oracle, nv, n_gl, n_sph):
    """Opt-in SURVEY 8(f1) mode (antipodal pairs merged, one forward FFT per radial-node segment) vs the oracle."""
    import bfsm
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    op = _make(bfsm, nv, n_gl, n_sph, exact=True)
    assert op.counters().antipodal_merged == 1 and op.counters().n_dirs == n_gl * n_sph // 2
    got = _collide(torch_cuda, op, f_h)
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max()
    op.destroy()


def test_exact_reductions_full_size_and_shards(torch_cuda):
    """cfg3 / cfg4 sizes: exact-reduction result == faithful result to rounding, also through 8 direction shards;
    and the published N=64, M_gl=64 BKW norms are reproduced in this mode too."""
    import bfsm
    torch = torch_cuda
    nv, n_gl = 64, 16
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    f = torch.from_numpy(f_h).cuda()
    for n_sph in (48, 156):
        a = _collide(torch, _make(bfsm, nv, n_gl, n_sph), f_h)
        b = _collide(torch, _make(bfsm, nv, n_gl, n_sph, exact=True), f_h)
        assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
        c = _sharded_collide(torch, bfsm, nv, n_gl, n_sph, f, 8, exact=True)
        assert np.abs(a - c).max() <= 1e-13 * np.abs(a).max()
    row = [r for r in GOLD["published"] if r["nv"] == 64 and r["n_sph"] == 12][0]
    fb, q_exact, _, dv = bfsm.bkw_solution(64)
    got = _collide(torch, _make(bfsm, 64, 64, 12, exact=True), fb)
    l1, l2, linf = bfsm.error_norms(got, q_exact, dv)
    assert abs(l2 - row["L2"]) <= 1e-10 and l2 == pytest.approx(row["L2"], rel=6e-9)
    assert linf == pytest.approx(row["Linf"], rel=6e-9)


def test_sharded_real_reduce_route(torch_cuda, oracle):
    """Default multi-GPU route: every shard inverse-transforms its own partial Q_gain_hat (bfsm_finish_partial, loss
    term on shard 0 only) and the real results are summed -- what the all-reduce of Q does on a multi-GPU node."""
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 32, 4, 12
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    f = torch.from_numpy(f_h).cuda()
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    for P in (2, 5):
        total = torch.zeros_like(f)
        for r in range(P):
            op = _make(bfsm, nv, n_gl, n_sph, shard=bfsm.shard_range(n_gl * n_sph, r, P))
            Qr = torch.empty_like(f)
            op.gainPartial(f)
            op.finishPartial(Qr, f, r == 0)
            op.synchronize()
            total += Qr
            op.destroy()
        got = total.cpu().numpy()
        assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max()


@pytest.mark.parametrize("gamma,b_gamma", [(1.0, 1.0 / (4.0 * np.pi)), (2.0, 0.3)])
def test_general_collision_kernels(torch_cuda, oracle, gamma, b_gamma):
    """hard spheres (gamma = 1) and gamma = 2 against the oracle, both modes, plus the discrete
    conservation sanity check that replaces the missing analytic answer: mass defect of Q is at truncation level."""
    import bfsm
    nv, n_gl, n_sph = 32, 8, 32
    c = bfsm.reference_constants()
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    ref = oracle.collide(f_h, oracle.gauss_legendre(n_gl, 0.0, c["R"]), oracle.spherical_design(n_sph), gamma, b_gamma, c["L"])
    for exact in (False, True):
        op = _make(bfsm, nv, n_gl, n_sph, exact=exact, gamma=gamma, b_gamma=b_gamma)
        got = _collide(torch_cuda, op, f_h)
        assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max()
        op.destroy()
    dv = 2 * c["L"] / nv
    assert abs(got.sum()) * dv ** 3 <= 5e-2 * np.abs(got).sum() * dv ** 3      # N=32, 8 radial nodes: truncation level


def test_bkw_relaxation_time_stepping(torch_cuda):
    """SURVEY 8(f3), caller side: SSP-RK3 on d f/dt = Q(f,f) with f resident on the device, from the BKW state at
    t = 5.5 to the reference's t = 6.5.  The result must track the exact BKW solution to the spectral accuracy of the
    grid, halve-dt must not change it (time error negligible), mass must be conserved to truncation level and the
    entropy must decrease towards the exact value."""
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 32, 16, 32
    op = _make(bfsm, nv, n_gl, n_sph, exact=True)
    a = bfsm.relax_bkw(op, nv, 5.5, 6.5, 10, torch)
    b = bfsm.relax_bkw(op, nv, 5.5, 6.5, 20, torch)
    op.destroy()
    assert a["l2_error"] < 2e-4 and b["l2_error"] < 2e-4              # f ~ 1e-2 .. 1e-1 at the origin
    assert abs(a["l2_error"] - b["l2_error"]) < 1e-6                   # RK3 error << spectral error
    assert a["mass_drift"] < 5e-4 and a["energy_drift"] < 5e-3
    assert a["entropy_end"] < a["entropy_start"]
    assert abs(a["entropy_end"] - a["entropy_exact_end"]) < 1e-3 * abs(a["entropy_exact_end"])


@pytest.mark.parametrize("nv,n_gl,n_sph,nb", [(16, 8, 32, 9), (32, 4, 12, 4), (64, 2, 12, 2)])
@pytest.mark.parametrize("exact", [False, True])
def test_batch_of_distributions(torch_cuda, oracle, nv, n_gl, n_sph, nb, exact):
    """bfsm_collide_batch == bfsm_collide member by member (bitwise) == oracle; a shorter batch than
    max_batch and an oversized one (rejected) are covered too."""
    import bfsm
    torch = torch_cuda
    f0 = bfsm.bkw_solution(nv)[0]
    fs_h = np.stack([bfsm.perturbed_input(f0, seed=100 + i, amp=0.05 * (i + 1)) for i in range(nb)])
    op = _make(bfsm, nv, n_gl, n_sph, exact=exact, max_batch=nb)
    fs = torch.from_numpy(fs_h).cuda()
    Qb = torch.empty_like(fs)
    op.computeCollisionBatch(Qb, fs, nb)
    Qb_h = Qb.cpu().numpy()
    single = torch.empty(nv ** 3, dtype=torch.float64, device="cuda")
    for i in (0, nb - 1):
        op(single, fs[i].reshape(-1).contiguous())
        assert np.array_equal(single.cpu().numpy().reshape(nv, nv, nv), Qb_h[i])
        ref = _oracle(oracle, fs_h[i], n_gl, n_sph)
        assert np.abs(Qb_h[i] - ref).max() <= TOL64 * np.abs(ref).max()
    Q2 = torch.empty(2 * nv ** 3, dtype=torch.float64, device="cuda") if nb > 2 else None
    if Q2 is not None:
        op.computeCollisionBatch(Q2, fs[:2].contiguous(), 2)
        assert np.array_equal(Q2.cpu().numpy().reshape(2, nv, nv, nv), Qb_h[:2])
    big = torch.zeros((nb + 1) * nv ** 3, dtype=torch.float64, device="cuda")
    with pytest.raises(bfsm.BfsmError):
        op.computeCollisionBatch(torch.empty_like(big), big, nb + 1)
    op.destroy()


def test_n128_fp32_matches_oracle(torch_cuda, oracle):
    """Config-5 grid (N=128, single precision, 192-point design is too slow for the CPU oracle: 12-point design,
    two radial nodes): faithful and exact-reduction paths against the fp64 oracle at fp32 tolerance."""
    import bfsm
    nv, n_gl, n_sph = 128, 2, 12
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    for exact in (False, True):
        op = _make(bfsm, nv, n_gl, n_sph, 32, exact=exact)
        got = _collide(torch_cuda, op, f_h)
        op.destroy()
        assert np.abs(got - ref).max() <= TOL32 * np.abs(ref).max()
    # the interleaved {A1', A2'} scratch of this geometry (faithful and exact modes; two arrays in the Hermitian mode) with
    # chunks, a batch of two (batch strides of the pairs and of P') and a pair of direction shards
    torch = torch_cuda
    fb = torch.from_numpy(np.stack([f_h, 0.5 * f_h])).cuda()
    for exact, herm in ((False, False), (True, False), (True, True)):
        op = _make(bfsm, nv, n_gl, n_sph, 32, exact=exact, hermitian=herm, max_chunk=7, max_batch=2)
        Qb = torch.empty_like(fb)
        op.computeCollisionBatch(Qb, fb, 2)
        torch.cuda.synchronize()
        op.destroy()
        assert np.abs(Qb[0].cpu().numpy() - ref).max() <= TOL32 * np.abs(ref).max(), (exact, herm)
        assert np.abs(Qb[1].cpu().numpy() - 0.25 * ref).max() <= TOL32 * np.abs(ref).max(), (exact, herm)
    f = torch.from_numpy(f_h).cuda()
    total = 0
    for r in range(2):
        op = _make(bfsm, nv, n_gl, n_sph, 32, shard=bfsm.shard_range(n_gl * n_sph, r, 2), max_chunk=5)
        Q = torch.empty_like(f)
        op.collidePartial(Q, f, r == 0)
        torch.cuda.synchronize()
        total = total + Q.cpu().numpy()
        op.destroy()
    assert np.abs(total - ref).max() <= TOL32 * np.abs(ref).max()


def test_n128_fp64_matches_oracle(torch_cuda, oracle):
    """N = 128 in double precision (the split-exchange geometry): faithful, exact-reduction and Hermitian paths against
    the oracle at the fp64 tolerance, and a two-member batch."""
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 128, 2, 12
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    for exact, herm in ((False, False), (True, False), (True, True)):
        op = _make(bfsm, nv, n_gl, n_sph, 64, exact=exact, hermitian=herm, max_chunk=7)
        got = _collide(torch, op, f_h)
        op.destroy()
        assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max(), (exact, herm)
    op = _make(bfsm, nv, n_gl, n_sph, 64, max_batch=2)
    fb = torch.from_numpy(np.stack([f_h, 0.5 * f_h])).cuda()
    Qb = torch.empty_like(fb)
    op.computeCollisionBatch(Qb, fb, 2)
    torch.cuda.synchronize()
    op.destroy()
    assert np.abs(Qb[0].cpu().numpy() - ref).max() <= TOL64 * np.abs(ref).max()
    assert np.abs(Qb[1].cpu().numpy() - 0.25 * ref).max() <= TOL64 * np.abs(ref).max()


def test_ka_cross_lane_build_matches_oracle(torch_cuda, oracle, tmp_path):
    """The cross-lane ("wavefront shuffle") form of KA's last line pass at N = 128 fp32 (csrc/bfsm_core.hpp ka_xlane,
    DevCtx::xlane_transpose8: v_permlane32_swap / v_permlane16_swap / DPP row_ror) is a build option (-DBFSM_KA_XLANE):
    measured 6 % slower than the LDS exchange in the kernel (profiles/r04_ka_xlane_lastpass_ab.txt), so it is not the
    default -- but it stays correct: built here and run in a child process against the oracle."""
    import subprocess
    import sys
    import bfsm
    root = os.path.dirname(HERE)
    pkg = os.path.join(root, "boltzmann-fourier-spectral-method_amd")
    lib = str(tmp_path / "libbfsm_xlane.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-function",
                           "-fno-slp-vectorize", "-DBFSM_KA_XLANE", "-shared", "-o", lib, os.path.join(pkg, "csrc", "bfsm_hip.hip")])
    nv, n_gl, n_sph = 128, 2, 12
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    np.save(tmp_path / "f.npy", f_h)
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {pkg!r})\n"
        "import bfsm\n"
        "c = bfsm.reference_constants()\n"
        f"f_h = np.load({str(tmp_path / 'f.npy')!r})\n"
        f"op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature({n_gl}, 0.0, c['R']), bfsm.SphericalDesign({n_sph}), {nv}, {nv}, {nv}, c['gamma'], c['b_gamma'], c['L'])\n"
        "op.setPrecision(32); op.setMaxChunk(5); op.initialize()\n"
        "f = torch.from_numpy(f_h).cuda(); Q = torch.empty_like(f)\n"
        "op(Q, f); torch.cuda.synchronize()\n"
        f"np.save({str(tmp_path / 'q.npy')!r}, Q.cpu().numpy())\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, BFSM_LIB=lib))
    assert out.returncode == 0, out.stderr[-2000:]
    got = np.load(tmp_path / "q.npy")
    assert np.abs(got - ref).max() <= TOL32 * np.abs(ref).max()


def test_cpp_multi_gpu_driver_on_one_device(torch_cuda):
    """host/maxwell_bkw_hip_multi.cpp = the reference driver with BoltzmannOperator<HIP_MultiGPU_Backend> (single
    process, f broadcast, direction shards, ONE grouped ncclReduce on Q).  On a one-GPU box: --gpus 1 without
    collectives, and --force-rccl (communicator of size 1: the broadcast / reduce calls of the P > 1 path are issued);
    cfg3 norms must match the golden values either way."""
    import re
    import subprocess
    pkg = os.path.join(os.path.dirname(HERE), "boltzmann-fourier-spectral-method_amd")
    exe = os.path.join(pkg, "maxwell_bkw_hip_multi")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", pkg, "-s", "maxwell_bkw_hip_multi"])
    for extra in ([], ["--force-rccl"], ["--force-rccl", "--hermitian"]):
        out = subprocess.run([exe, "--Nv", "64", "--Ngl", "16", "--Ns", "48", "-t", "3", "--gpus", "1",
                              "--design-dir", os.path.join(pkg, "data", "sph_design")] + extra,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        row = [r for r in GOLD["survey"] if r["nv"] == 64][0]
        got = {k: float(re.search(k + r" error: (\S+)", out.stdout).group(1)) for k in ("L1", "L2", "Linf")}
        for k in ("L1", "L2", "Linf"):
            assert got[k] == pytest.approx(row[k], rel=6e-9), (k, got, extra)     # 9 printed digits
            assert re.fullmatch(r"\d\.\d{8}e[-+]\d\d", re.search(k + r" error: (\S+)", out.stdout).group(1))
    # BASELINE config 5's form of the driver (single precision, its N = 128 grid; one radial node here), chunked, an explicit
    # device list, the collectives forced on, per-device counters: the error norm against the analytic
    # BKW collision term equals the fp64 run's to fp32 rounding
    out = subprocess.run([exe, "--Nv", "128", "--Ngl", "2", "--Ns", "192", "-t", "2", "--devices", "0", "--precision", "32",
                          "--chunk", "100", "--force-rccl", "--counters", "--design-dir", os.path.join(pkg, "data", "sph_design")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "precision = 32" in out.stdout and re.search(r"device 0: directions 384, chunks 4 x 96, kernels \S+ ms", out.stdout), out.stdout
    assert '"precision": 32' in out.stdout
    l2_32 = float(re.search(r"L2 error: (\S+)", out.stdout).group(1))
    out = subprocess.run([exe, "--Nv", "128", "--Ngl", "2", "--Ns", "192", "-t", "1", "--gpus", "1",
                          "--design-dir", os.path.join(pkg, "data", "sph_design")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    l2_64 = float(re.search(r"L2 error: (\S+)", out.stdout).group(1))
    # two radial nodes: a quadrature error of a few per cent that both precisions share to fp32 rounding
    assert abs(l2_32 - l2_64) <= 5e-6 * l2_64, (l2_32, l2_64)
    # --input random and bad flags
    out = subprocess.run([exe, "--Nv", "32", "--Ngl", "4", "--Ns", "12", "-t", "1", "--input", "random",
                          "--design-dir", os.path.join(pkg, "data", "sph_design")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "sum |Q| =" in out.stdout
    for bad in (["--devices", "0,0"], ["--devices", "7,x"], ["--precision", "16"], ["--gpus", "2", "--devices", "0"]):
        out = subprocess.run([exe, "--Nv", "32", "--Ngl", "4", "--Ns", "12"] + bad, capture_output=True, text=True, timeout=120)
        assert out.returncode != 0 and "error:" in out.stderr, bad


class _UserQuadrature:
    """A caller's own SphericalQuadrature (Quadratures/AbstractSphericalQuadratures.hpp:21-42): 13 nodes -- the 12-point
    design rotated by a generic rotation plus one more node -- with unequal weights: no antipodal structure, no
    permutation symmetry, an odd count."""

    def __init__(self, oracle):
        x, y, z, _ = oracle.spherical_design(12)
        a, b, c = 0.37, 1.13, -0.61                          # Euler angles of a generic rotation
        ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
        R = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1]]) @ np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]]) @ \
            np.array([[1, 0, 0], [0, cc, -sc], [0, sc, cc]])
        pts = np.vstack([(R @ np.vstack([x, y, z])).T, [[0.6, -0.48, 0.64]]])
        self._x, self._y, self._z = (np.ascontiguousarray(pts[:, k]) for k in range(3))
        w = np.linspace(0.8, 1.3, 13)
        self._w = w * (4 * np.pi / w.sum())

    def getx(self): return self._x
    def gety(self): return self._y
    def getz(self): return self._z
    def getWeights(self): return self._w
    def getNumberOfPoints(self): return 13
    def as_tuple(self): return (self._x, self._y, self._z, self._w)


@pytest.mark.parametrize("mode", ["faithful", "exact", "hermitian"])
@pytest.mark.parametrize("nv,n_gl", [(32, 3), (64, 2)])
def test_user_supplied_quadrature_through_the_c_abi(torch_cuda, oracle, nv, n_gl, mode):
    """The C-ABI takes arbitrary sx / sy / sz / sph_wts arrays like the reference's abstract SphericalQuadrature; every other
    GPU case uses a shipped symmetric design with equal weights.  Here: 13 rotated, non-antipodal nodes with unequal weights,
    whole field on the perturbed input against the oracle, faithful and with the exact-reduction flags (nothing can be
    merged: antipodal_merged == 0, only the linearity part applies), plus a pair of direction shards."""
    import bfsm
    torch = torch_cuda
    c = bfsm.reference_constants()
    sph = _UserQuadrature(oracle)
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    ref = oracle.collide(f_h, oracle.gauss_legendre(n_gl, 0.0, c["R"]), sph.as_tuple(), c["gamma"], c["b_gamma"], c["L"])

    def make(shard=None):
        op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), sph, nv, nv, nv, c["gamma"], c["b_gamma"], c["L"])
        op.setExactReductions(mode != "faithful", hermitian=(mode == "hermitian"))
        if shard:
            op.setDirectionShard(*shard)
        op.initialize()
        return op

    op = make()
    cn = op.counters()
    assert cn.antipodal_merged == 0 and cn.exact_reductions == (0 if mode == "faithful" else 1) and cn.n_dirs == 13 * n_gl
    got = _collide(torch, op, f_h)
    op.destroy()
    assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max()
    f = torch.from_numpy(f_h).cuda()
    total = 0
    for r in range(2):                                       # shards that cut through a radial node's 13 directions
        op = make(bfsm.shard_range(13 * n_gl, r, 2))
        Q = torch.empty_like(f)
        op.collidePartial(Q, f, r == 0)
        torch.cuda.synchronize()
        total = total + Q.cpu().numpy()
        op.destroy()
    assert np.abs(total - ref).max() <= TOL64 * np.abs(ref).max()


def test_cpp_multi_gpu_operator_batches_and_input_stream(torch_cuda, tmp_path):
    """tests/host/test_multigpu_hip.cpp: BoltzmannOperator<HIP_MultiGPU_Backend> on real HIP + RCCL (collectives forced on;
    two devices where the box has them): a batch equals its single evaluations bitwise, f produced on a non-blocking
    stream is waited for through setInputStream(), counters() per device, setMaxChunk() reaches the devices."""
    import subprocess
    root = os.path.dirname(HERE)
    pkg = os.path.join(root, "boltzmann-fourier-spectral-method_amd")
    exe = str(tmp_path / "test_multigpu_hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(pkg, "host"),
                           "-I", os.path.join(root, "include"), "-x", "hip", os.path.join(root, "tests", "host", "test_multigpu_hip.cpp"),
                           os.path.join(pkg, "host", "HIPMultiGPUBoltzmannOperator.cpp"),
                           os.path.join(pkg, "host", "HIPBoltzmannOperator.cpp"), os.path.join(pkg, "host", "Quadratures", "SphericalDesign.cpp"),
                           "-L" + pkg, "-lbfsm_hip", "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib",
                           "-o", exe])
    out = subprocess.run([exe, os.path.join(pkg, "data", "sph_design")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "multi-GPU operator checks passed" in out.stdout


@pytest.mark.parametrize("nv,n_gl,n_sph", [(16, 8, 32), (32, 8, 48), (64, 2, 12)])
def test_hermitian_reduction_matches_oracle(torch_cuda, oracle, nv, n_gl, n_sph):
    """BFSM_FLAG_HERMITIAN (planes lx = 0..N/2 only + exact Nyquist terms) against the oracle on an input with energy
    in all Nyquist planes; also the fp32 variant at N = 128."""
    import bfsm
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    op = _make(bfsm, nv, n_gl, n_sph, hermitian=True)
    got = _collide(torch_cuda, op, f_h)
    op.destroy()
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max()


def test_hermitian_full_size(torch_cuda, oracle):
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 64, 16, 48
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    a = _collide(torch, _make(bfsm, nv, n_gl, n_sph), f_h)
    b = _collide(torch, _make(bfsm, nv, n_gl, n_sph, hermitian=True), f_h)
    assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    f128 = bfsm.perturbed_input(bfsm.bkw_solution(128)[0])
    ref = _oracle(oracle, f128, 2, 12)
    got = _collide(torch, _make(bfsm, 128, 2, 12, 32, hermitian=True), f128)
    assert np.abs(got - ref).max() <= TOL32 * np.abs(ref).max()


def test_bench_two_ranks_rehearsal(torch_cuda):
    """`python bench.py --gpus 2` invoked DIRECTLY (no launcher): bench.py starts its own two ranks.  A one-GPU box
    cannot host two RCCL ranks, so the rehearsal backend (gloo, both ranks on device 0) is used: same self-launch, same
    sharding, same bfsm.sharded_step on the HIP operator, same timing and JSON code path as the 8-GPU run; only the
    collective's transport differs."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BFSM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "cfg2", "--no-roofline"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["directions_per_gpu"] == 192 and d["cpu_baseline"] is None
    assert d["config"]["directions_per_gpu_min"] == 192 and d["config"]["directions_per_gpu_max"] == 192
    assert d["config"]["rank_devices"] == [0, 0]          # the gloo rehearsal shares device 0; RCCL runs refuse that
    assert d["config"]["collective"]["ranks"] == 2 and d["config"]["collective_overlap"] is False
    assert d["blocking_call"]["value"] > 0 and d["overlapped"]["value"] > 0
    assert d["exact_reductions"]["value"] > 0


def test_bench_refuses_more_ranks_than_devices(torch_cuda):
    """With the RCCL backend every rank needs its own GPU: asking for more is an error (rc 4), not a silent sharing."""
    import subprocess
    import sys
    import torch
    root = os.path.dirname(HERE)
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BFSM_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode != 0 and "device(s) visible" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_two_real_gpus_when_present(torch_cuda, oracle):
    """On a box with >= 2 GPUs (the development box has one: skipped there): the single-process multi-GPU operator
    (maxwell_bkw_hip_multi --gpus 2: ncclCommInitAll, broadcast, per-device threads, ncclReduce) reproduces the cfg3
    golden norms, and bench.py --gpus 2 over RCCL prints its line."""
    import re
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(HERE)
    pkg = os.path.join(root, "boltzmann-fourier-spectral-method_amd")
    out = subprocess.run([os.path.join(pkg, "maxwell_bkw_hip_multi"), "--Nv", "64", "--Ngl", "16", "--Ns", "48", "-t", "3",
                          "--gpus", "2", "--design-dir", os.path.join(pkg, "data", "sph_design")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    row = [r for r in GOLD["survey"] if r["nv"] == 64][0]
    for k in ("L1", "L2", "Linf"):
        assert float(re.search(k + r" error: (\S+)", out.stdout).group(1)) == pytest.approx(row[k], rel=6e-9)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BFSM_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["collective"]["ranks"] == 2 and "nccl" in d["config"]["collective"]["backend"]


_FULL_REF = {}


def _full_ref(oracle, nv, n_gl, n_sph):
    """Oracle field on the perturbed (non-symmetric, Nyquist-populated) input, computed once per configuration."""
    import bfsm
    key = (nv, n_gl, n_sph)
    if key not in _FULL_REF:
        f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
        _FULL_REF[key] = (f_h, _oracle(oracle, f_h, n_gl, n_sph))
    return _FULL_REF[key]


@pytest.mark.parametrize("mode", ["faithful", "exact", "hermitian"])
@pytest.mark.parametrize("nv,n_gl,n_sph", [(64, 16, 48), (64, 16, 156), (128, 2, 48)],
                         ids=["cfg3", "cfg4", "N128_2x48"])
def test_full_size_field_matches_oracle(torch_cuda, oracle, nv, n_gl, n_sph, mode):
    """The whole fp64 field at BASELINE.json's config 3 and config 4 sizes (and N = 128 with 96 directions) against the
    oracle on an input without any symmetry: max|Q - Q_oracle| <= 1e-12 max|Q_oracle|, in all three modes (the
    isotropic BKW norms cannot see a symmetry shortcut; this can)."""
    import bfsm
    f_h, ref = _full_ref(oracle, nv, n_gl, n_sph)
    op = _make(bfsm, nv, n_gl, n_sph, 64, exact=(mode != "faithful"), hermitian=(mode == "hermitian"))
    got = _collide(torch_cuda, op, f_h)
    op.destroy()
    assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max()


@pytest.mark.parametrize("mode", ["faithful", "exact", "hermitian"])
def test_full_size_field_fp32_n128_matches_oracle(torch_cuda, oracle, mode):
    """The single-precision variant of config 5's grid on the same 2 x 48-direction quadrature and the same cached
    oracle field as the fp64 case above (96 directions instead of the 24 of test_n128_fp32_matches_oracle), whole
    field, all three modes, at the fp32 tolerance."""
    import bfsm
    f_h, ref = _full_ref(oracle, 128, 2, 48)
    op = _make(bfsm, 128, 2, 48, 32, exact=(mode != "faithful"), hermitian=(mode == "hermitian"))
    got = _collide(torch_cuda, op, f_h)
    op.destroy()
    assert np.abs(got - ref).max() <= TOL32 * np.abs(ref).max()


def test_failed_create_does_not_poison_the_next_handle(torch_cuda):
    """A create that fails with BFSM_ERR_NOMEM (scratch for an absurd batch) must not leave a sticky HIP error behind
    that the next, healthy handle's first launch would be blamed for: the documented recovery is to retry smaller."""
    import bfsm
    from bfsm import capi
    torch = torch_cuda
    with pytest.raises(bfsm.BfsmError) as e:
        _make(bfsm, 64, 16, 48, max_batch=400)           # 2 x 400 x 768 x 4 MiB of A1'/A2' scratch: 2.4 TB
    assert e.value.code == 4                              # BFSM_ERR_NOMEM
    op = _make(bfsm, 16, 2, 6)
    f = torch.from_numpy(bfsm.bkw_solution(16)[0]).cuda()
    Q = torch.empty_like(f)
    op(Q, f)                                              # raises on any non-zero status
    assert torch.isfinite(Q).all()
    op.destroy()


def test_cpp_relaxation_driver(torch_cuda):
    """host/bkw_relax_hip.cpp: the C++ time-stepping caller (SSP-RK3, f resident on the device) tracks the exact BKW
    solution at N=32 to the grid's spectral accuracy and conserves mass to truncation level."""
    import re
    import subprocess
    pkg = os.path.join(os.path.dirname(HERE), "boltzmann-fourier-spectral-method_amd")
    exe = os.path.join(pkg, "bkw_relax_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", pkg, "-s", "bkw_relax_hip"])
    out = subprocess.run([exe, "--Nv", "32", "--Ngl", "16", "--Ns", "32", "--steps", "10", "--exact-reductions",
                          "--design-dir", os.path.join(pkg, "data", "sph_design")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    l2 = float(re.search(r"L2 error vs exact BKW: (\S+)", out.stdout).group(1))
    mass = float(re.search(r"relative mass drift: (\S+)", out.stdout).group(1))
    h = re.search(r"entropy: (\S+) -> (\S+)", out.stdout)
    assert l2 < 2e-4 and mass < 5e-4 and float(h.group(2)) < float(h.group(1))


def test_hip_matches_committed_q_fixtures(torch_cuda):
    """HIP path against the committed data fixtures (config 1: N=16, M_gl=8, 32-point design), all three modes."""
    import bfsm
    f0 = bfsm.bkw_solution(16)[0]
    for name, f_h in (("q_cfg1_bkw.npy", f0), ("q_cfg1_random.npy", bfsm.perturbed_input(f0))):
        want = np.load(os.path.join(HERE, "golden", name))
        for kw in ({}, {"exact": True}, {"hermitian": True}):
            op = _make(bfsm, 16, 8, 32, **kw)
            got = _collide(torch_cuda, op, f_h)
            op.destroy()
            assert np.abs(got - want).max() <= TOL64 * np.abs(want).max(), (name, kw)


@pytest.mark.parametrize("exact", [False, True])
def test_evaluation_is_graph_capturable(torch_cuda, exact):
    """The async entry points only enqueue kernels (no allocation, no synchronisation, no attribute calls after the
    first evaluation), so a caller can capture an evaluation in a HIP graph and replay it on new contents of f."""
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 32, 4, 12
    op = _make(bfsm, nv, n_gl, n_sph, exact=exact, hermitian=exact)
    f0 = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    f = torch.from_numpy(f0).cuda()
    Q, Qg = torch.empty_like(f), torch.zeros_like(f)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        op.computeCollisionAsync(Qg, f, side.cuda_stream)          # first call outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        op.computeCollisionAsync(Qg, f, torch.cuda.current_stream().cuda_stream)
    for scale in (1.0, 0.5):
        f.copy_(torch.from_numpy(f0 * scale))
        Qg.zero_()
        g.replay()
        torch.cuda.synchronize()
        op(Q, f)
        assert torch.equal(Q, Qg)
    op.destroy()


def test_cpp_fft_benchmark_driver(torch_cuda):
    """host/fft_benchmark_hip.cpp (format of the reference's cufft_benchmark.cu over bfsm_fft3d): the batched
    forward + inverse round trip of the all-ones array is exact and the report has the reference's shape."""
    import re
    import subprocess
    pkg = os.path.join(os.path.dirname(HERE), "boltzmann-fourier-spectral-method_amd")
    exe = os.path.join(pkg, "fft_benchmark_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", pkg, "-s", "fft_benchmark_hip"])
    for extra in ([], ["--precision", "32"]):
        out = subprocess.run([exe, "--Nv", "16", "--Ns", "6", "-t", "2"] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert "Total number of samples taken: 2" in out.stdout and "Run statistics for HIP" in out.stdout
        assert float(re.search(r"L1 error: (\S+)", out.stdout).group(1)) <= (1e-10 if not extra else 1e-2)


@pytest.mark.parametrize("nv,n_gl,n_sph,exact", [(16, 8, 32, False), (64, 2, 12, False), (32, 4, 12, True)])
def test_fused_collide_is_bitwise_the_two_call_sequence(torch_cuda, nv, n_gl, n_sph, exact):
    """bfsm_collide and bfsm_collide_partial_async (slab reduce fused into the tail) against bfsm_gain_partial +
    bfsm_finish / bfsm_finish_partial on the same handle: same bits.  (N = 16: with the whole-direction kernels switched
    off -- they are a different summation order, compared with the oracle in their own test.)"""
    import bfsm
    torch = torch_cuda
    f = torch.from_numpy(bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])).cuda()
    op = _make(bfsm, nv, n_gl, n_sph, exact=exact, hermitian=exact, max_chunk=7, small_path=False)
    Qa, Qb = torch.empty_like(f), torch.empty_like(f)
    op(Qa, f)                                   # fused
    op.gainPartial(f)
    op.finish(Qb, f)
    torch.cuda.synchronize()
    assert torch.equal(Qa, Qb)
    op.destroy()
    B = n_gl * n_sph
    shard = _make(bfsm, nv, n_gl, n_sph, shard=(B // 3, B), exact=exact, hermitian=exact, small_path=False)
    for with_loss in (False, True):
        shard.collidePartial(Qa, f, with_loss)
        shard.gainPartial(f)
        shard.finishPartial(Qb, f, with_loss)
        torch.cuda.synchronize()
        assert torch.equal(Qa, Qb)
    shard.destroy()


def test_error_paths_through_the_c_abi(torch_cuda):
    """Status codes + messages, never an exit or an exception across the boundary (include/bfsm.h conventions)."""
    import ctypes
    import bfsm
    from bfsm import capi
    torch = torch_cuda
    L = capi.load_library()
    op = _make(bfsm, 16, 2, 6, max_batch=2)
    h = op._h
    f = torch.zeros(2 * 16 ** 3, dtype=torch.float64, device="cuda")
    Q = torch.empty_like(f)
    null = ctypes.c_void_p(0)
    assert L.bfsm_collide(h, null, ctypes.c_void_p(f.data_ptr())) == 1                       # BFSM_ERR_INVALID
    assert b"null" in L.bfsm_last_error(h)
    assert L.bfsm_collide_batch(h, ctypes.c_void_p(Q.data_ptr()), ctypes.c_void_p(f.data_ptr()), 3) == 1
    assert b"n_batch" in L.bfsm_last_error(h)
    assert L.bfsm_collide_batch(h, ctypes.c_void_p(Q.data_ptr()), ctypes.c_void_p(f.data_ptr()), 0) == 1
    assert L.bfsm_fft3d(h, ctypes.c_void_p(f.data_ptr()), 1, 2) == 1                         # sign must be +-1
    assert L.bfsm_fft3d(h, null, 1, 1) == 1
    assert L.bfsm_collide(None, ctypes.c_void_p(Q.data_ptr()), ctypes.c_void_p(f.data_ptr())) == 1
    assert L.bfsm_synchronize(None) == 1 and L.bfsm_destroy(None) == 0
    # the handle is still usable after rejected calls
    assert L.bfsm_collide_batch(h, ctypes.c_void_p(Q.data_ptr()), ctypes.c_void_p(f.data_ptr()), 2) == 0
    assert torch.isfinite(Q).all()
    op.destroy()
    # a descriptor asking for a device that does not exist
    one = np.ones(4)
    dp = ctypes.POINTER(ctypes.c_double)
    p = one.ctypes.data_as(dp)
    bad = capi.Desc(16, 16, 16, 4, 4, p, p, p, p, p, p, 0.0, 1.0, 1.0, 64, 99, 0, 0, 0, 0)
    hh = ctypes.c_void_p()
    assert L.bfsm_create(ctypes.byref(bad), ctypes.byref(hh)) == 1 and not hh.value
    assert b"device" in L.bfsm_last_error(None)


def test_randomised_plans_against_the_oracle(torch_cuda, oracle):
    """Seeded sweep on the hardware: grid size, quadrature sizes, direction shard, chunk size, mode and batch size vary;
    every case is compared with the oracle (whole evaluations) or with the sum rule of shards."""
    import bfsm
    torch = torch_cuda
    rng = np.random.default_rng(4242)
    fs = {nv: bfsm.perturbed_input(bfsm.bkw_solution(nv)[0]) for nv in (16, 32)}
    for case in range(30):
        nv = int(rng.choice([16, 32]))
        n_gl = int(rng.integers(1, 6))
        n_sph = int(rng.choice([6, 12, 32, 48]))
        B = n_gl * n_sph
        mode = int(rng.integers(0, 3))
        exact, herm = mode >= 1, mode == 2
        max_chunk = int(rng.choice([0, 1, 3, 7, 13, 50]))
        f_h = fs[nv]
        ref = _oracle(oracle, f_h, n_gl, n_sph)
        tag = (case, nv, n_gl, n_sph, mode, max_chunk)
        if rng.random() < 0.5:
            op = _make(bfsm, nv, n_gl, n_sph, exact=exact, hermitian=herm, max_chunk=max_chunk)
            got = _collide(torch, op, f_h)
            op.destroy()
        else:                                   # two or three shards, default (real-Q) combination
            cuts = sorted(set([0, B] + [int(c) for c in rng.integers(0, B + 1, size=2)]))
            f = torch.from_numpy(f_h).cuda()
            got = np.zeros_like(f_h)
            for i, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
                op = _make(bfsm, nv, n_gl, n_sph, shard=(lo, hi), exact=exact, hermitian=herm, max_chunk=max_chunk)
                Qr = torch.empty_like(f)
                op.collidePartial(Qr, f, i == 0)
                torch.cuda.synchronize()
                got += Qr.cpu().numpy()
                op.destroy()
        assert np.abs(got - ref).max() <= TOL64 * np.abs(ref).max(), tag


def test_full_config5_against_the_analytic_bkw_collision_term(torch_cuda):
    """Config 5 at full size (N=128, M_gl=30, 192-point design, B=5760): the spectral method is converged there, so
    the fp64 path must reproduce the analytic BKW collision term to rounding level, and the single-precision variant
    must agree with it to fp32 rounding (known-answer test at BASELINE.json's largest size, both precisions)."""
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 128, 30, 192
    f_h, q_exact, _, dv = bfsm.bkw_solution(nv)
    f = torch.from_numpy(f_h).cuda()
    out = {}
    for prec in (64, 32):
        op = _make(bfsm, nv, n_gl, n_sph, prec)
        Q = torch.empty_like(f)
        op(Q, f)
        out[prec] = Q.cpu().numpy()
        op.destroy()
    l2 = float(np.sqrt(((out[64] - q_exact) ** 2).sum() * dv ** 3))
    assert l2 <= 5e-15, l2                                                    # measured 3.3e-16
    assert np.abs(out[32] - out[64]).max() <= 1e-5 * np.abs(out[64]).max()      # measured 1.1e-6


def _make_box(bfsm, shape, n_gl, n_sph, precision=64, shard=None, max_chunk=0, gamma=0.0, b_gamma=1.0 / (4.0 * np.pi), L=11.0,
              max_batch=0):
    c = bfsm.reference_constants()
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph),
                                   shape[0], shape[1], shape[2], gamma, b_gamma, L)
    op.setPrecision(precision)
    if shard:
        op.setDirectionShard(*shard)
    if max_chunk:
        op.setMaxChunk(max_chunk)
    op.setMaxBatch(max_batch)
    op.initialize()
    return op


@pytest.mark.parametrize("shape,n_gl,n_sph,max_chunk", [((32, 64, 16), 4, 12, 0), ((48, 48, 48), 4, 12, 0),
                                                        ((96, 96, 96), 2, 6, 0), ((64, 48, 80), 2, 12, 5),
                                                        ((16, 128, 32), 3, 6, 0), ((20, 36, 50), 3, 12, 7),
                                                        ((28, 22, 26), 2, 6, 0), ((112, 14, 8), 2, 6, 0),
                                                        # fused sequence, groups of 4 directions across radial nodes
                                                        ((96, 8, 10), 6, 6, 0),
                                                        # x-line kernel between plane kernels / between per-axis passes
                                                        ((128, 8, 8), 5, 6, 0), ((16, 28, 12), 2, 6, 0)])
def test_any_box_matches_oracle(torch_cuda, oracle, shape, n_gl, n_sph, max_chunk):
    """Grid generality of the reference's constructors (CUDABoltzmannOperator.hpp:48-54; plans cu:86-100): non-cubic
    boxes and sizes with factors 3 and 5 match the oracle at the fp64 tolerance (the cubes 48^3 and 96^3 on the fused
    pipeline's radix-3 geometries, everything else on the size-generic path)."""
    import bfsm
    torch = torch_cuda
    c = bfsm.reference_constants()
    rng = np.random.default_rng(sum(shape))
    f_h = rng.random(shape) + 0.1
    op = _make_box(bfsm, shape, n_gl, n_sph, max_chunk=max_chunk, gamma=0.5, b_gamma=0.3)
    f = torch.from_numpy(f_h).cuda()
    Q = torch.empty_like(f)
    op(Q, f)
    op.destroy()
    ref = oracle.collide(f_h, oracle.gauss_legendre(n_gl, 0.0, c["R"]), oracle.spherical_design(n_sph), 0.5, 0.3, 11.0)
    assert np.abs(Q.cpu().numpy() - ref).max() <= TOL64 * np.abs(ref).max()


def test_any_box_shards_batches_fp32_and_fft(torch_cuda, oracle):
    """The rest of the C-ABI on a non-cubic box: direction shards summed through the handle-owned buffer, a batch of
    distributions, the single-precision variant and bfsm_fft3d (natural layouts on this path)."""
    import bfsm
    torch = torch_cuda
    c = bfsm.reference_constants()
    shape, n_gl, n_sph = (32, 48, 16), 3, 12
    rng = np.random.default_rng(5)
    f_h = rng.random(shape) + 0.1
    ref = oracle.collide(f_h, oracle.gauss_legendre(n_gl, 0.0, c["R"]), oracle.spherical_design(n_sph), 0.0, 1.0 / (4.0 * np.pi), 11.0)
    f = torch.from_numpy(f_h).cuda()
    # two shards, partial Q_gain_hat summed, tail on shard 0
    ops = [_make_box(bfsm, shape, n_gl, n_sph, shard=bfsm.shard_range(n_gl * n_sph, r, 2)) for r in range(2)]
    views = []
    for op in ops:
        op.gainPartial(f)
        op.synchronize()
        views.append(_DevView(*op.qhatBuffer()).tensor(torch))
    views[0].add_(views[1])
    Q = torch.empty_like(f)
    torch.cuda.synchronize()
    ops[0].finish(Q, f)
    ops[0].synchronize()
    assert np.abs(Q.cpu().numpy() - ref).max() <= TOL64 * np.abs(ref).max()
    # real-space route: each shard transforms its own partial sum
    Qa, Qb = torch.empty_like(f), torch.empty_like(f)
    ops[0].collidePartial(Qa, f, True)
    ops[1].collidePartial(Qb, f, False)
    torch.cuda.synchronize()
    assert np.abs((Qa + Qb).cpu().numpy() - ref).max() <= TOL64 * np.abs(ref).max()
    for op in ops:
        op.destroy()
    # batch of two
    op = _make_box(bfsm, shape, n_gl, n_sph, max_batch=2)
    fb = torch.from_numpy(np.stack([f_h, 0.5 * f_h])).cuda()
    Qb2 = torch.empty_like(fb)
    op.computeCollisionBatch(Qb2, fb, 2)
    torch.cuda.synchronize()
    assert np.abs(Qb2[0].cpu().numpy() - ref).max() <= TOL64 * np.abs(ref).max()
    assert np.abs(Qb2[1].cpu().numpy() - 0.25 * ref).max() <= TOL64 * np.abs(ref).max()
    # batch x direction shard (what a device of the multi-GPU operator runs): members go through the launches together,
    # each rank's partial results add up to the whole, the loss term rides on rank 0; chunks of 7 directions, fp32 as well
    for prec, tol in ((64, TOL64), (32, TOL32)):
        parts = []
        for r in range(2):
            o2 = _make_box(bfsm, shape, n_gl, n_sph, precision=prec, shard=bfsm.shard_range(n_gl * n_sph, r, 2), max_chunk=7, max_batch=2)
            Qp = torch.empty_like(fb)
            o2.collideBatchPartial(Qp, fb, 2, r == 0)
            torch.cuda.synchronize()
            parts.append(Qp.cpu().numpy())
            o2.destroy()
        tot = parts[0] + parts[1]
        assert np.abs(tot[0] - ref).max() <= tol * np.abs(ref).max()
        assert np.abs(tot[1] - 0.25 * ref).max() <= tol * np.abs(ref).max()
    # transforms
    a = rng.standard_normal((2,) + shape) + 1j * rng.standard_normal((2,) + shape)
    d = torch.from_numpy(a).cuda()
    op.fft3d(d, 2, -1)
    fw = np.fft.fftn(a, axes=(1, 2, 3))
    assert np.abs(d.cpu().numpy() - fw).max() <= 4e-15 * np.abs(fw).max()
    op.fft3d(d, 2, +1)
    assert np.abs(d.cpu().numpy() / a[0].size - a).max() <= 1e-14 * np.abs(a).max()
    op.destroy()
    # fp32
    op = _make_box(bfsm, shape, n_gl, n_sph, precision=32)
    Q32 = torch.empty_like(f)
    op(Q32, f)
    op.destroy()
    assert np.abs(Q32.cpu().numpy() - ref).max() <= TOL32 * np.abs(ref).max()


def test_cpp_driver_on_a_size_with_factor_three(torch_cuda):
    """maxwell_bkw_hip --Nv 48: the reference's driver takes any --Nv (maxwell_bkw_cuda.cu:31); the BKW error at N = 48
    must sit between the published N = 32 and N = 64 values (spectral convergence)."""
    import re
    import subprocess
    pkg = os.path.join(os.path.dirname(HERE), "boltzmann-fourier-spectral-method_amd")
    out = subprocess.run([os.path.join(pkg, "maxwell_bkw_hip"), "--Nv", "48", "--Ns", "12", "-t", "2", "--design-dir",
                          os.path.join(pkg, "data", "sph_design")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    l2 = float(re.search(r"L2 error: (\S+)", out.stdout).group(1))
    assert GOLD["published"][2]["L2"] < l2 < GOLD["published"][0]["L2"]


@pytest.mark.parametrize("mode", ["faithful", "exact", "hermitian"])
@pytest.mark.parametrize("n_gl,n_sph,prec", [(8, 32, 64), (40, 12, 64), (3, 6, 64), (8, 32, 32)])
def test_n16_whole_direction_kernels_and_tile_pipeline_agree_with_oracle(torch_cuda, oracle, n_gl, n_sph, prec, mode):
    """N = 16 single evaluations run on the whole-direction kernels (two launches); BFSM_FLAG_NO_SMALL_PATH sends the
    same call through the plane-tile pipeline.  Both against the oracle, all three modes, plus a shard without loss."""
    import bfsm
    torch = torch_cuda
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(16)[0])
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    # fp32 at N = 16: Q is the small difference of gain and loss (the grid does not resolve f), so rounding relative to
    # max|Q| is amplified: measured 1.3e-5 (whole-direction kernels) / 8.6e-6 (tile pipeline) on config 1
    tol = TOL64 if prec == 64 else 5e-5
    for small in (True, False):
        op = _make(bfsm, 16, n_gl, n_sph, prec, exact=(mode != "faithful"), hermitian=(mode == "hermitian"), small_path=small)
        got = _collide(torch, op, f_h)
        q2 = _collide(torch, op, f_h)
        op.destroy()
        assert np.array_equal(got, q2)                                   # deterministic (no atomics)
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), small
    B = n_gl * n_sph
    f = torch.from_numpy(f_h).cuda()
    parts = []
    for r in range(2):
        op = _make(bfsm, 16, n_gl, n_sph, prec, shard=bfsm.shard_range(B, r, 2), exact=(mode != "faithful"),
                   hermitian=(mode == "hermitian"))
        Q = torch.empty_like(f)
        op.collidePartial(Q, f, r == 0)
        torch.cuda.synchronize()
        parts.append(Q.cpu().numpy())
        op.destroy()
    assert np.abs(parts[0] + parts[1] - ref).max() <= tol * np.abs(ref).max()


@pytest.mark.parametrize("nv,n_gl,n_sph,nb", [(32, 4, 12, 3), (16, 8, 32, 4), (64, 2, 12, 2)])
def test_batch_times_direction_shards(torch_cuda, oracle, nv, n_gl, n_sph, nb):
    """SURVEY 8(f4) x 8(e) composed: every rank evaluates its direction shard for the whole batch in one call
    (bfsm_collide_batch_partial_async), ONE sum over the ranks gives every member's Q."""
    import bfsm
    torch = torch_cuda
    f0 = bfsm.bkw_solution(nv)[0]
    fs_h = np.stack([bfsm.perturbed_input(f0, seed=7 + i, amp=0.03 * (i + 1)) for i in range(nb)])
    fs = torch.from_numpy(fs_h).cuda()
    B, P = n_gl * n_sph, 3
    total = torch.zeros_like(fs)
    for r in range(P):
        op = _make(bfsm, nv, n_gl, n_sph, shard=bfsm.shard_range(B, r, P), max_batch=nb)
        Q = torch.empty_like(fs)
        op.collideBatchPartial(Q, fs, nb, r == 0)
        torch.cuda.synchronize()
        total += Q                                   # what the all-reduce of the batch does
        op.destroy()
    got = total.cpu().numpy()
    for i in range(nb):
        ref = _oracle(oracle, fs_h[i], n_gl, n_sph)
        assert np.abs(got[i] - ref).max() <= TOL64 * np.abs(ref).max()


def test_synchronize_waits_for_every_stream_the_handle_used(torch_cuda, oracle):
    """include/bfsm.h: bfsm_synchronize waits for every stream that was passed to the handle since the last
    synchronize, not only the most recent one.  Two evaluations are enqueued on two different streams (ordered by an
    event, as the header requires for calls that share the scratch); after ONE bfsm_synchronize both results are
    complete without any torch-side synchronisation."""
    import bfsm
    torch = torch_cuda
    nv, n_gl, n_sph = 64, 4, 48
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    op = _make(bfsm, nv, n_gl, n_sph)
    f = torch.from_numpy(f_h).cuda()
    f2 = 2.0 * f                           # produced on torch's current stream ...
    Qa, Qb = torch.zeros_like(f), torch.zeros_like(f)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()               # ... and complete before s1 / s2 read it
    op.computeCollisionAsync(Qa, f, s1.cuda_stream)
    ev = torch.cuda.Event()
    ev.record(s1)
    s2.wait_event(ev)                      # same handle, same scratch: the second call is ordered behind the first
    op.computeCollisionAsync(Qb, f2, s2.cuda_stream)
    op.synchronize()                       # must cover s1 AND s2
    a, b = Qa.cpu().numpy(), Qb.cpu().numpy()
    assert np.abs(a - ref).max() <= TOL64 * np.abs(ref).max()
    assert np.abs(b - 4.0 * ref).max() <= TOL64 * np.abs(4.0 * ref).max()
    op.destroy()


def _hip_runtime():
    """libamdhip64 through ctypes: raw streams that can really be destroyed (torch's streams come from a pool)."""
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    rt.hipStreamCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    rt.hipStreamCreate.restype = ctypes.c_int
    rt.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    rt.hipStreamDestroy.restype = ctypes.c_int
    rt.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    rt.hipStreamSynchronize.restype = ctypes.c_int
    return rt


def test_a_destroyed_stream_does_not_break_the_handle(torch_cuda, oracle):
    """include/bfsm.h: the caller's stream handle is never used after an entry point returns.  A side stream is
    destroyed between an async call and bfsm_synchronize (with and without the caller's own wait in between): the
    synchronize returns BFSM_OK, the result is complete, and the next blocking call on the handle works."""
    import ctypes
    import bfsm
    torch = torch_cuda
    rt = _hip_runtime()
    nv, n_gl, n_sph = 32, 4, 12
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    op = _make(bfsm, nv, n_gl, n_sph)
    f = torch.from_numpy(f_h).cuda()
    torch.cuda.synchronize()
    for wait_first in (True, False):
        Q = torch.zeros_like(f)
        torch.cuda.synchronize()
        s = ctypes.c_void_p()
        assert rt.hipStreamCreate(ctypes.byref(s)) == 0
        op.computeCollisionAsync(Q, f, s.value)
        if wait_first:
            assert rt.hipStreamSynchronize(s) == 0
        assert rt.hipStreamDestroy(s) == 0          # HIP lets queued work finish; the handle value is dead from here on
        op.synchronize()                            # BFSM_OK (raises otherwise): waits on the handle's own event
        assert np.abs(Q.cpu().numpy() - ref).max() <= TOL64 * np.abs(ref).max()
        Q2 = torch.zeros_like(f)
        op(Q2, f)                                   # blocking call: bfsm_collide -> bfsm_synchronize
        assert np.array_equal(Q2.cpu().numpy(), Q.cpu().numpy())
    op.destroy()


def test_async_calls_on_many_streams_never_block(torch_cuda, oracle):
    """More distinct streams than the handle tracks (64) in a row, each destroyed right after its call: no entry point
    fails or synchronises a dead stream, and a final bfsm_synchronize returns BFSM_OK."""
    import ctypes
    import bfsm
    torch = torch_cuda
    rt = _hip_runtime()
    nv, n_gl, n_sph = 16, 2, 6
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    ref = _oracle(oracle, f_h, n_gl, n_sph)
    op = _make(bfsm, nv, n_gl, n_sph)
    f = torch.from_numpy(f_h).cuda()
    Q = torch.zeros_like(f)
    torch.cuda.synchronize()
    live = []
    for i in range(80):
        s = ctypes.c_void_p()
        assert rt.hipStreamCreate(ctypes.byref(s)) == 0
        live.append(s)                              # 80 distinct handle values alive at once
    for s in live:
        op.computeCollisionAsync(Q, f, s.value)
        assert rt.hipStreamSynchronize(s) == 0      # calls on one handle must be ordered: here by the host
    for s in live:
        assert rt.hipStreamDestroy(s) == 0
    op.synchronize()
    assert np.abs(Q.cpu().numpy() - ref).max() <= TOL64 * np.abs(ref).max()
    op.destroy()


def test_batch_of_one_equals_a_single_evaluation_bitwise(torch_cuda):
    """include/bfsm.h: a batch member is bitwise the single evaluation on the same handle -- including the N = 16
    single-evaluation handle, whose bfsm_collide runs on the whole-direction kernels (a batch of one takes them too)."""
    import bfsm
    torch = torch_cuda
    for nv, n_gl, n_sph in ((16, 4, 12), (32, 2, 12)):
        f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
        f = torch.from_numpy(f_h).cuda().reshape(-1)
        op = _make(bfsm, nv, n_gl, n_sph)
        Q1, Qb = torch.empty_like(f), torch.empty_like(f)
        op(Q1, f)
        op.computeCollisionBatch(Qb, f, 1)
        assert np.array_equal(Q1.cpu().numpy(), Qb.cpu().numpy())
        op.destroy()


@pytest.mark.parametrize("mode", ["faithful", "exact", "hermitian"])
@pytest.mark.parametrize("nv,n_gl,n_sph,prec", [(48, 4, 12, 64), (96, 2, 12, 64), (48, 4, 12, 32), (96, 2, 12, 32),
                                                (80, 2, 12, 64), (80, 2, 12, 32), (24, 4, 12, 64), (40, 4, 12, 64),
                                                (24, 4, 12, 32), (40, 4, 12, 32)])
def test_fused_radix3_sizes_match_oracle(torch_cuda, oracle, nv, n_gl, n_sph, prec, mode):
    """N = 48, 96 (prime-factor 4 x 3 / 8 x 3 register transforms, 4 threads per line), N = 80 (4 x 5) and N = 24 (two
    threads per line) on the fused three-kernel pipeline: whole field on the perturbed input against the oracle, all
    three modes, both precisions; plus a direction shard pair."""
    import bfsm
    torch = torch_cuda
    f_h, ref = _full_ref(oracle, nv, n_gl, n_sph)
    tol = TOL64 if prec == 64 else TOL32
    op = _make(bfsm, nv, n_gl, n_sph, prec, exact=(mode != "faithful"), hermitian=(mode == "hermitian"))
    assert op.counters().exact_reductions == (0 if mode == "faithful" else 1)      # the fused pipeline, not the generic path
    got = _collide(torch, op, f_h)
    op.destroy()
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    if mode == "faithful" and prec == 64:
        B = n_gl * n_sph
        f = torch.from_numpy(f_h).cuda()
        parts = []
        for r in range(2):
            op = _make(bfsm, nv, n_gl, n_sph, prec, shard=bfsm.shard_range(B, r, 2))
            Q = torch.empty_like(f)
            op.collidePartial(Q, f, r == 0)
            torch.cuda.synchronize()
            parts.append(Q.cpu().numpy())
            op.destroy()
        assert np.abs(parts[0] + parts[1] - ref).max() <= tol * np.abs(ref).max()


def test_size_generic_path_accepts_and_ignores_the_reduction_flags(torch_cuda, oracle):
    """include/bfsm.h: on boxes served by the size-generic path BFSM_FLAG_EXACT_REDUCTIONS / BFSM_FLAG_HERMITIAN are
    accepted and have no effect: same result bit for bit, and the counters report exact_reductions = 0."""
    import bfsm
    torch = torch_cuda
    c = bfsm.reference_constants()
    shape, n_gl, n_sph = (16, 24, 8), 2, 12
    f_h = np.random.default_rng(3).random(shape) + 0.1
    f = torch.from_numpy(f_h).cuda()
    res = []
    for exact, herm in ((False, False), (True, False), (True, True)):
        op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph),
                                       shape[0], shape[1], shape[2], 0.0, 1.0 / (4.0 * np.pi), 11.0)
        op.setExactReductions(exact, hermitian=herm)
        op.initialize()
        cn = op.counters()
        assert cn.exact_reductions == 0 and cn.antipodal_merged == 0
        Q = torch.empty_like(f)
        op(Q, f)
        res.append(Q.cpu().numpy())
        op.destroy()
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])
    ref = oracle.collide(f_h, oracle.gauss_legendre(n_gl, 0.0, c["R"]), oracle.spherical_design(n_sph), 0.0, 1.0 / (4.0 * np.pi), 11.0)
    assert np.abs(res[0] - ref).max() <= TOL64 * np.abs(ref).max()
