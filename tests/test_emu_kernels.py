"""CPU unit tests of the HIP kernel bodies through the host lock-step emulator (tests/emu).

The emulator runs the exact device code of csrc/bfsm_core.hpp (every GPU thread a coroutine, __syncthreads a
yield) under the exact plan + launch sequence of csrc/bfsm_pipeline.hpp, so layouts, LDS exchange addresses,
twiddles, chunking and slab bookkeeping are checked against the oracle without a GPU.  The -m gpu tests then only
have to confirm that the hardware executes the same code.
"""
import numpy as np
import pytest

import emu_lib as E

GAMMA, B_GAMMA, R = 0.0, 1.0 / (4.0 * np.pi), 10.0


@pytest.mark.parametrize("n,prec,tol", [(16, 64, 2e-15), (32, 64, 2e-15), (64, 64, 3e-15), (16, 32, 1e-6),
                                        (32, 32, 1e-6), (64, 32, 1e-6), (128, 32, 1e-6), (128, 64, 3e-15),
                                        (48, 64, 3e-15), (96, 64, 3e-15), (48, 32, 1e-6), (96, 32, 1e-6),
                                        (24, 64, 3e-15), (80, 64, 3e-15), (80, 32, 1e-6)])
def test_fft3d_matches_numpy_and_round_trips(n, prec, tol):
    """Mirrors the reference's FFT check (fftw_benchmark.cpp:137-171): forward, scale 1/G, inverse."""
    rng = np.random.default_rng(n + prec)
    batch = 1 if n >= 64 else 2
    a = rng.standard_normal((batch, n, n, n)) + 1j * rng.standard_normal((batch, n, n, n))
    ref = np.fft.fftn(a, axes=(1, 2, 3))
    fw = E.fft3d(a, -1, prec)
    assert np.abs(fw - ref).max() <= tol * np.abs(ref).max()
    back = E.fft3d(ref, +1, prec) / n ** 3
    assert np.abs(back - a).max() <= tol * np.abs(a).max()


@pytest.mark.parametrize("nv,n_gl,n_sph,max_chunk,prec,tol", [
    (16, 2, 6, 0, 64, 1e-12), (16, 3, 12, 5, 64, 1e-12), (32, 2, 6, 4, 64, 1e-12), (16, 2, 6, 0, 32, 2e-5),
    (48, 2, 6, 5, 64, 1e-12), (48, 1, 6, 0, 32, 1e-4), (96, 1, 6, 0, 64, 1e-12), (80, 1, 6, 0, 64, 1e-12), (24, 3, 12, 5, 64, 1e-12), (40, 2, 6, 0, 64, 1e-12),      # radix-3 geometries (E = 12 / 24, T = 4); fp32: Q = gain - loss cancels on a 6-direction rule
])
def test_collide_matches_oracle(oracle, nv, n_gl, n_sph, max_chunk, prec, tol):
    f, _, L, _ = oracle.bkw(nv)
    f = oracle.perturbed_input(f)          # non-symmetric: exercises the Nyquist planes
    gl = oracle.gauss_legendre(n_gl, 0.0, R)
    sph = oracle.spherical_design(n_sph)
    Q, qhat = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, prec, max_chunk=max_chunk)
    Qo, qo = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    assert np.abs(qhat - qo).max() <= tol * np.abs(qo).max()
    assert np.abs(Q - Qo).max() <= tol * np.abs(Qo).max()


@pytest.mark.slow
def test_collide_n64_matches_oracle(oracle):
    f, _, L, _ = oracle.bkw(64)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(2, 0.0, R)
    sph = oracle.spherical_design(6)
    Q, qhat = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, max_chunk=4)
    Qo, qo = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    assert np.abs(qhat - qo).max() <= 1e-12 * np.abs(qo).max()
    assert np.abs(Q - Qo).max() <= 1e-12 * np.abs(Qo).max()


@pytest.mark.slow
def test_collide_n128_fp64_matches_oracle(oracle):
    """The split-exchange geometry (N = 128 in fp64: real and imaginary parts exchanged one after the other, f_hat
    re-read per direction, segment sum formed before the (y,z) forward transform)."""
    f, _, L, _ = oracle.bkw(128)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(1, 0.0, R)
    sph = oracle.spherical_design(6)
    Q, qhat = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, max_chunk=4)
    Qo, qo = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    assert np.abs(qhat - qo).max() <= 1e-12 * np.abs(qo).max()
    assert np.abs(Q - Qo).max() <= 1e-12 * np.abs(Qo).max()


@pytest.mark.slow
def test_collide_n128_fp32_pipelined_pair_matches_oracle(oracle):
    """N = 128 in single precision: KA processes the two signs of a direction as a software-pipelined pair of tiles
    through one exchange buffer (12 barriers per direction).  The emulator advances the waves adversarially, so a
    missing barrier between a read phase of one tile and the write phase of the other reads stale LDS here (6-point
    design, one radial node: every workgroup column loops over several directions, so the hand-over of the exchange
    buffer from one direction to the next is covered as well)."""
    f, _, L, _ = oracle.bkw(128)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(1, 0.0, R)
    sph = oracle.spherical_design(6)
    # two chunks of three directions: the interleaved {A1', A2'} scratch and the P' buffer of this geometry are re-used by
    # the second chunk with its own direction offset (round 4)
    Q, qhat = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 32, max_chunk=4)
    Qo, qo = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    assert np.abs(qhat - qo).max() <= 2e-5 * np.abs(qo).max()           # fp32 rounding: measured 7.6e-6


def test_direction_shards_add_up(oracle):
    """Partial Q_gain_hat of disjoint shards (what each GPU contributes to the reduce) sums to the whole."""
    f, _, L, _ = oracle.bkw(16)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(3, 0.0, R)
    sph = oracle.spherical_design(12)
    _, whole = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    parts = 0
    for rng_ in ((0, 7), (7, 20), (20, 36)):
        _, qh = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, dir_range=rng_, max_chunk=5, want_Q=False)
        po = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, dir_range=rng_, return_qhat=True)[1]
        assert np.abs(qh - po).max() <= 1e-12 * np.abs(whole).max()
        parts = parts + qh
    assert np.abs(parts - whole).max() <= 1e-12 * np.abs(whole).max()


def test_empty_shard_gives_zero_gain(oracle):
    f, _, L, _ = oracle.bkw(16)
    gl = oracle.gauss_legendre(2, 0.0, R)
    sph = oracle.spherical_design(6)
    _, qh = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, dir_range=(5, 5), want_Q=False)
    assert np.all(qh == 0)


def test_plan_chunks_and_segments_cover_shard_once():
    """Host logic: chunks respect max_chunk (default: whole shard up to 1024 directions) and tile the shard; the
    accumulation segments tile every chunk, never straddle a radial node (beta1 is applied per slab), and give
    every x-plane at least 8 workgroups when the chunk has that many directions."""
    for nv, n_gl, n_sph, rng_, mc in [(64, 16, 48, (0, 0), 0), (64, 16, 156, (0, 0), 0), (128, 30, 192, (0, 0), 0),
                                      (64, 16, 156, (312, 624), 0), (16, 3, 12, (7, 20), 5), (32, 8, 48, (0, 0), 7),
                                      (64, 2, 12, (0, 0), 0), (64, 1, 6, (0, 0), 0), (64, 16, 48, (100, 101), 0)]:
        prec = 32 if nv == 128 else 64
        chunks, segs = E.plan(nv, n_gl, n_sph, prec, rng_, mc)
        b0, b1 = rng_ if rng_ != (0, 0) else (0, n_gl * n_sph)
        cap = mc or 1024
        assert len(chunks) == -(-(b1 - b0) // cap)
        pos, seg_pos = 0, 0
        for ci, (n_seg, d0, n, per_group, seg0) in enumerate(chunks):
            assert d0 == pos and 1 <= n <= cap and seg0 == seg_pos
            groups = ((512 if nv >= 64 else 2048 if nv == 32 else 1024) + nv - 1) // nv
            assert per_group * groups >= n
            inner = 0
            for (c, sd0, sn, r) in segs[seg0:seg0 + n_seg]:
                assert c == ci and sd0 == inner and sn >= 1
                g_first, g_last = b0 + d0 + sd0, b0 + d0 + sd0 + sn - 1
                assert g_first // n_sph == r == g_last // n_sph
                inner += sn
            assert inner == n
            assert n_seg >= min(n, groups)
            pos += n
            seg_pos += n_seg
        assert pos == b1 - b0 and seg_pos == len(segs)


def test_plan_rejects_unsupported():
    with pytest.raises(ValueError):
        E.plan(50 * 7, 2, 6)                      # 350: beyond 256 and a factor 7
    with pytest.raises(ValueError):
        E.plan(34, 2, 6)                          # prime factor 17
    E.plan(128, 2, 6, 64)                # N=128 in fp64 is planned like any other size (split-exchange tiles)
    with pytest.raises(ValueError):
        E.plan(16, 2, 6, 16)             # precision must be 32 or 64
    with pytest.raises(ValueError):
        E.plan(16, 2, 6, 64, (5, 40))    # shard beyond n_gl*n_sph


EXACT = 2   # BFSM_FLAG_EXACT_REDUCTIONS


@pytest.mark.parametrize("nv,n_gl,n_sph,max_chunk", [(16, 3, 12, 0), (16, 3, 12, 5), (32, 2, 6, 0), (48, 2, 6, 0)])
def test_exact_reductions_match_oracle(oracle, nv, n_gl, n_sph, max_chunk):
    """antipodal pairs merged + one forward FFT per radial-node segment == all directions, to rounding."""
    f, _, L, _ = oracle.bkw(nv)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(n_gl, 0.0, R)
    sph = oracle.spherical_design(n_sph)
    Qo, qo = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    Q, qhat = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, max_chunk=max_chunk, flags=EXACT)
    assert np.abs(qhat - qo).max() <= 1e-12 * np.abs(qo).max()
    assert np.abs(Q - Qo).max() <= 1e-12 * np.abs(Qo).max()
    chunks, segs = E.plan(nv, n_gl, n_sph, 64, (0, 0), max_chunk, flags=EXACT, sph=sph)
    assert sum(c[2] for c in chunks) == n_gl * n_sph // 2          # half the directions are evaluated


def test_exact_reductions_without_antipodal_structure(oracle):
    """A rule that is not antipodal (pairs broken by a permutation, unequal weights) only gets the linearity part."""
    f, _, L, _ = oracle.bkw(16)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(2, 0.0, R)
    x, y, z, w = oracle.spherical_design(12)
    perm = np.array([0, 7, 2, 3, 4, 5, 6, 1, 8, 9, 10, 11])
    sph = (x[perm].copy(), y[perm].copy(), z[perm].copy(), w * np.linspace(0.9, 1.1, 12))
    Qo, qo = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    Q, qhat = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, flags=EXACT)
    assert np.abs(qhat - qo).max() <= 1e-12 * np.abs(qo).max()
    assert np.abs(Q - Qo).max() <= 1e-12 * np.abs(Qo).max()
    chunks, _ = E.plan(16, 2, 12, 64, flags=EXACT, sph=sph)
    assert sum(c[2] for c in chunks) == 24                         # nothing merged


def test_exact_reductions_sharded(oracle):
    f, _, L, _ = oracle.bkw(16)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(3, 0.0, R)
    sph = oracle.spherical_design(12)
    _, whole = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    parts = 0
    for rng_ in ((0, 7), (7, 20), (20, 36)):      # full-direction shards map to proportional effective ranges
        _, qh = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, dir_range=rng_, want_Q=False, flags=EXACT)
        parts = parts + qh
    assert np.abs(parts - whole).max() <= 1e-12 * np.abs(whole).max()


@pytest.mark.parametrize("gamma,b_gamma", [(1.0, 1.0 / (4.0 * np.pi)), (2.0, 0.3), (-0.5, 0.1)])
def test_general_collision_kernels(oracle, gamma, b_gamma):
    """gamma != 0 (hard spheres gamma = 1, ...) only changes the tabulated weights rho^(gamma+2)
    (FFTWBoltzmannOperator.cpp:252,290-293); the operator is checked against the oracle for those too."""
    f, _, L, _ = oracle.bkw(16)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(3, 0.0, R)
    sph = oracle.spherical_design(12)
    Qo = oracle.collide(f, gl, sph, gamma, b_gamma, L)
    for flags in (0, EXACT):
        Q, _ = E.collide(f, gl, sph, gamma, b_gamma, L, 64, flags=flags)
        assert np.abs(Q - Qo).max() <= 1e-12 * np.abs(Qo).max()


@pytest.mark.parametrize("flags", [0, EXACT])
def test_batch_of_distributions(oracle, flags):
    """n_batch distributions through one set of launches == the same distributions one at a time
    (bitwise), and == the oracle."""
    f0, _, L, _ = oracle.bkw(16)
    fs = np.stack([oracle.perturbed_input(f0, seed=s, amp=0.1 + 0.05 * i) for i, s in enumerate((1, 2, 3))])
    gl = oracle.gauss_legendre(3, 0.0, R)
    sph = oracle.spherical_design(12)
    Qb = E.collide_batch(fs, gl, sph, GAMMA, B_GAMMA, L, max_chunk=7, flags=flags)
    for i in range(3):
        Qi, _ = E.collide(fs[i], gl, sph, GAMMA, B_GAMMA, L, 64, max_chunk=7, flags=flags)
        assert np.array_equal(Qb[i], Qi)
        Qo = oracle.collide(fs[i], gl, sph, GAMMA, B_GAMMA, L)
        assert np.abs(Qb[i] - Qo).max() <= 1e-12 * np.abs(Qo).max()


HERMITIAN = 4   # BFSM_FLAG_HERMITIAN (only together with EXACT)


@pytest.mark.parametrize("nv,n_gl,n_sph,max_chunk", [(16, 3, 12, 0), (16, 3, 12, 5), (32, 2, 6, 0), (48, 2, 6, 0),
                                                     (96, 1, 6, 0), (80, 1, 6, 0), (24, 3, 12, 0), (40, 2, 6, 0),
                                                     (64, 2, 6, 0), (64, 3, 12, 4)])   # N = 64: KN rides in KA's grid
def test_hermitian_reduction_matches_oracle(oracle, nv, n_gl, n_sph, max_chunk):
    """f real => A'[-lx] = conj A'[lx] + exact rank-one Nyquist terms: only the planes lx = 0..N/2 are computed and
    stored.  The perturbed input has energy in all three Nyquist planes, so a wrong correction shows at 1e-3."""
    f, _, L, _ = oracle.bkw(nv)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(n_gl, 0.0, R)
    sph = oracle.spherical_design(n_sph)
    Qo, qo = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
    Q, qhat = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, max_chunk=max_chunk, flags=EXACT | HERMITIAN)
    assert np.abs(qhat - qo).max() <= 1e-12 * np.abs(qo).max()
    assert np.abs(Q - Qo).max() <= 1e-12 * np.abs(Qo).max()


def test_hermitian_n64_batch_with_guest_workgroups(oracle):
    """N = 64: the Nyquist-row workgroups are appended to KA's grid (nyq_rides_along); with a batch the grid has a z
    extent as well and every member must still equal its single evaluation bitwise and the oracle to 1e-12."""
    f0, _, L, _ = oracle.bkw(64)
    fs = np.stack([oracle.perturbed_input(f0, seed=s) for s in (3, 4)])
    gl = oracle.gauss_legendre(2, 0.0, R)
    sph = oracle.spherical_design(6)
    Qb = E.collide_batch(fs, gl, sph, GAMMA, B_GAMMA, L, max_chunk=2, flags=EXACT | HERMITIAN)
    for i in range(2):
        Qi, _ = E.collide(fs[i], gl, sph, GAMMA, B_GAMMA, L, 64, max_chunk=2, flags=EXACT | HERMITIAN)
        assert np.array_equal(Qb[i], Qi)
        Qo = oracle.collide(fs[i], gl, sph, GAMMA, B_GAMMA, L)
        assert np.abs(Qb[i] - Qo).max() <= 1e-12 * np.abs(Qo).max()


def test_hermitian_needs_exact_flag():
    with pytest.raises(ValueError):
        E.plan(16, 2, 6, 64, flags=HERMITIAN)


def test_hermitian_batch_and_shards(oracle):
    f0, _, L, _ = oracle.bkw(16)
    fs = np.stack([oracle.perturbed_input(f0, seed=s) for s in (5, 6)])
    gl = oracle.gauss_legendre(3, 0.0, R)
    sph = oracle.spherical_design(12)
    Qb = E.collide_batch(fs, gl, sph, GAMMA, B_GAMMA, L, max_chunk=7, flags=EXACT | HERMITIAN)
    for i in range(2):
        Qo, whole = oracle.collide(fs[i], gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)
        assert np.abs(Qb[i] - Qo).max() <= 1e-12 * np.abs(Qo).max()
    parts = 0
    for rng_ in ((0, 13), (13, 36)):
        _, qh = E.collide(fs[1], gl, sph, GAMMA, B_GAMMA, L, 64, dir_range=rng_, want_Q=False, flags=EXACT | HERMITIAN)
        parts = parts + qh
    assert np.abs(parts - whole).max() <= 1e-12 * np.abs(whole).max()


@pytest.mark.parametrize("nv,n_gl,n_sph,flags", [(16, 3, 12, 0), (32, 2, 6, 0), (16, 3, 12, EXACT)])
def test_fused_reduce_tail_is_bitwise_the_two_call_sequence(oracle, nv, n_gl, n_sph, flags):
    """bfsm_collide / bfsm_collide_batch / bfsm_collide_partial_async fuse the slab reduce into the first tail kernel;
    the sum keeps the order of the reduce kernel, so Q is bitwise what gain_partial + finish give.  (N = 16: with the
    whole-direction kernels off -- they are a different summation order and have their own test.)"""
    flags |= 8                                # BFSM_FLAG_NO_SMALL_PATH
    f, _, L, _ = oracle.bkw(nv)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(n_gl, 0.0, R)
    sph = oracle.spherical_design(n_sph)
    Q2, _ = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, max_chunk=5, flags=flags)            # two calls
    Q1 = E.collide_batch(f[None], gl, sph, GAMMA, B_GAMMA, L, 64, max_chunk=5, flags=flags)[0]   # fused
    assert np.array_equal(Q1, Q2)


def test_randomised_plans_against_the_oracle(oracle):
    """Seeded sweep over small shapes the hand-picked cases do not hit: single radial node, shards that start and end
    inside a radial node, chunk sizes that divide nothing, every mode -- each shard's partial Q_gain_hat against the
    oracle's for the same direction range (plan / segment / slab bookkeeping is what varies here)."""
    rng = np.random.default_rng(20261004)
    f, _, L, _ = oracle.bkw(16)
    f = oracle.perturbed_input(f)
    for case in range(36):
        n_gl = int(rng.integers(1, 5))
        n_sph = int(rng.choice([6, 12, 32]))
        B = n_gl * n_sph
        flags = int(rng.choice([0, EXACT, EXACT | HERMITIAN]))
        max_chunk = int(rng.choice([0, 1, 3, 5, 7, 11]))
        if rng.random() < 0.5:
            lo = int(rng.integers(0, B))
            hi = int(rng.integers(lo, B + 1))
        else:
            lo, hi = 0, B
        if flags and (lo, hi) != (0, B):
            # merged (antipodal) directions: a shard is given in full directions and maps proportionally; compare the
            # sum of a two-way split with the whole instead of a direction range of the oracle
            mid = int(rng.integers(0, B + 1))
            gl = oracle.gauss_legendre(n_gl, 0.0, R)
            sph = oracle.spherical_design(n_sph)
            whole = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)[1]
            parts = sum(E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, dir_range=r, max_chunk=max_chunk, want_Q=False,
                                  flags=flags)[1] for r in ((0, mid), (mid, B)) if r[0] != r[1])
            assert np.abs(parts - whole).max() <= 1e-12 * np.abs(whole).max(), (case, n_gl, n_sph, flags, max_chunk, mid)
            continue
        gl = oracle.gauss_legendre(n_gl, 0.0, R)
        sph = oracle.spherical_design(n_sph)
        rng_ = (lo, hi) if (lo, hi) != (0, B) else (0, 0)
        want_Q = rng_ == (0, 0)
        Q, qh = E.collide(f, gl, sph, GAMMA, B_GAMMA, L, 64, dir_range=rng_, max_chunk=max_chunk, want_Q=want_Q, flags=flags)
        if lo == hi:
            assert np.all(qh == 0)
            continue
        ref = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, dir_range=(lo, hi), return_qhat=True)
        scale = np.abs(oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, return_qhat=True)[1]).max()
        assert np.abs(qh - ref[1]).max() <= 1e-12 * scale, (case, n_gl, n_sph, flags, max_chunk, lo, hi)
        if want_Q:
            assert np.abs(Q - ref[0]).max() <= 1e-12 * np.abs(ref[0]).max(), (case, n_gl, n_sph, flags, max_chunk)


@pytest.mark.parametrize("shape,n_gl,n_sph,prec,tol,kw", [
    ((8, 4, 12), 3, 12, 64, 1e-12, {}),                      # non-cubic, radix 2 / 4 / 3
    ((12, 12, 12), 2, 6, 64, 1e-12, {}),                     # cubic but not a fused size (radix 3)
    ((20, 20, 20), 1, 6, 64, 1e-12, {}),                     # cubic, radix 5, not a fused size
    ((20, 10, 4), 2, 6, 64, 1e-12, {}),                      # radix 5
    ((16, 8, 6), 3, 12, 64, 1e-12, {"max_chunk": 5}),        # several chunks of directions
    ((16, 8, 6), 3, 12, 64, 1e-12, {"dir_range": (7, 29)}),  # a direction shard
    ((16, 8, 6), 5, 6, 64, 1e-12, {}),                       # fused sequence: groups of 15 directions across radial nodes
    ((12, 6, 10), 4, 6, 32, 2e-5, {"max_chunk": 17}),        # ... single precision, ragged chunks
    ((8, 14, 6), 2, 6, 64, 1e-12, {}),                       # x-line kernel between per-axis passes (a radix-7 y axis: no plane kernel)
    ((100, 4, 6), 1, 6, 64, 1e-12, {}),                      # 100-point x lines (three 16-line buffers still fit)
    ((160, 4, 6), 1, 6, 64, 1e-12, {}),                      # long x lines: 8 lines per workgroup in the x passes and the x-line kernel
    ((4, 14, 160), 1, 6, 64, 1e-12, {}),                     # ... in a z pass (no plane kernel: radix-7 y axis)
    ((154, 4, 4), 1, 6, 64, 1e-12, {}),                      # ... with the table-driven radices (154 = 2 x 7 x 11)
    ((8, 16, 4), 2, 6, 32, 2e-5, {}),                        # single-precision variant
    ((14, 22, 26), 2, 6, 64, 1e-12, {}),                     # radices 7, 11, 13 (table-driven butterflies)
    ((28, 6, 4), 2, 6, 64, 1e-12, {}),                       # 7 behind 4
])
def test_size_generic_path_matches_oracle(oracle, shape, n_gl, n_sph, prec, tol, kw):
    """Grids outside the fused pipeline's cubes (csrc/bfsm_generic.hpp: one mixed-radix Stockham pass per axis, pointwise
    steps fused on the load side) against the oracle, which handles any box like the reference does
    (FFTWBoltzmannOperator.cpp:64-65 plans Nvx x Nvy x Nvz)."""
    rng = np.random.default_rng(sum(shape))
    f = rng.random(shape) + 0.1                              # no symmetry at all
    gl = oracle.gauss_legendre(n_gl, 0.0, R)
    sph = oracle.spherical_design(n_sph)
    L = 11.0
    rng_ = kw.get("dir_range")
    Q, qhat = E.collide(f, gl, sph, 0.5, 0.3, L, prec, max_chunk=kw.get("max_chunk", 0), dir_range=rng_ or (0, 0),
                        want_Q=rng_ is None)
    Qo, qo = oracle.collide(f, gl, sph, 0.5, 0.3, L, dir_range=rng_, return_qhat=True)
    assert np.abs(qhat - qo).max() <= tol * np.abs(qo).max()
    if rng_ is None:
        assert np.abs(Q - Qo).max() <= tol * np.abs(Qo).max()


@pytest.mark.parametrize("shape,max_chunk", [((16, 8, 6), 0), ((16, 8, 6), 7), ((8, 14, 6), 0)])
def test_size_generic_batch_of_distributions(oracle, shape, max_chunk):
    """A batch on the size-generic path: the fused sequence takes all members through every launch (own f_hat, scratch,
    slabs and Q_hat per member), the other sequences one member after the other -- either way bitwise the single
    evaluations, and the oracle's result."""
    rng = np.random.default_rng(11)
    fs = rng.random((3,) + shape) + 0.1
    gl = oracle.gauss_legendre(3, 0.0, R)
    sph = oracle.spherical_design(6)
    Qb = E.collide_batch(fs, gl, sph, 0.5, 0.3, 11.0, max_chunk=max_chunk)
    for i in range(3):
        Qi, _ = E.collide(fs[i], gl, sph, 0.5, 0.3, 11.0, 64, max_chunk=max_chunk)
        assert np.array_equal(Qb[i], Qi)
        Qo = oracle.collide(fs[i], gl, sph, 0.5, 0.3, 11.0)
        assert np.abs(Qb[i] - Qo).max() <= 1e-12 * np.abs(Qo).max()


@pytest.mark.parametrize("n_gl,n_sph,flags,rng,prec,tol", [
    (3, 12, 0, (0, 0), 64, 1e-12),          # faithful: one forward transform per direction
    (8, 32, 0, (0, 0), 64, 1e-12),          # config 1: 256 directions, one per workgroup
    (40, 12, 0, (0, 0), 64, 1e-12),         # 480 directions: two per workgroup, runs cross radial nodes
    (3, 12, 2, (0, 0), 64, 1e-12),          # exact reductions: antipodal pairs merged, products summed per radial run
    (3, 12, 6, (0, 0), 64, 1e-12),          # + hermitian flag (same kernels on this path)
    (3, 12, 0, (5, 29), 64, 1e-12),         # a direction shard
    (2, 6, 0, (0, 0), 32, 2e-5),            # single precision
])
def test_whole_direction_kernels_n16(oracle, n_gl, n_sph, flags, rng, prec, tol):
    """N = 16: a direction lives in one workgroup's registers + LDS (small_gain / small_reduce / small_tail, three
    launches per evaluation) -- the sequence bfsm_collide runs at this size -- against the oracle."""
    f, _, L, _ = oracle.bkw(16)
    f = oracle.perturbed_input(f)
    gl = oracle.gauss_legendre(n_gl, 0.0, R)
    sph = oracle.spherical_design(n_sph)
    got = E.collide_partial(f, gl, sph, GAMMA, B_GAMMA, L, prec, dir_range=rng, flags=flags)
    ref = oracle.collide(f, gl, sph, GAMMA, B_GAMMA, L, dir_range=rng if rng != (0, 0) else None)
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    # the same call with the whole-direction kernels switched off takes the plane-tile pipeline: same answer
    other = E.collide_partial(f, gl, sph, GAMMA, B_GAMMA, L, prec, dir_range=rng, flags=flags | 8)
    assert np.abs(other - ref).max() <= tol * np.abs(ref).max()
    if rng != (0, 0):       # a rank that does not own the loss term
        g = E.collide_partial(f, gl, sph, GAMMA, B_GAMMA, L, prec, dir_range=rng, with_loss=False, flags=flags)
        g2 = E.collide_partial(f, gl, sph, GAMMA, B_GAMMA, L, prec, dir_range=rng, with_loss=False, flags=flags | 8)
        assert np.abs(g - g2).max() <= tol * np.abs(ref).max()


def test_reciprocal_division_of_the_size_generic_passes_is_exact():
    """csrc/bfsm_generic.hpp gen_div: floor(w / d) = (int)((float(w) + 0.5f) * (1.0f / d)) for every w < 65536, d <= 256 -- the
    index arithmetic of the size-generic pass loops.  Checked exhaustively in float32, also with the reciprocal one ulp
    low and one ulp high (a device reciprocal that is not correctly rounded)."""
    w = np.arange(65536, dtype=np.int64)
    wf = w.astype(np.float32) + np.float32(0.5)
    for d in range(1, 257):
        inv = np.float32(1.0) / np.float32(d)
        for r in (inv, np.nextafter(inv, np.float32(0)), np.nextafter(inv, np.float32(2))):
            got = (wf * r).astype(np.int32)            # float32 product, truncation: what the kernels do
            assert np.array_equal(got, w // d), d
