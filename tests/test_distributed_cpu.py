"""World-size-2 (and 3) gloo tests of the N>1 path: contiguous direction shards, ONE all-reduce of the partial
Q_gain_hat, replicated tail -- the same bfsm.sharded_step() that bench.py runs over RCCL, with the host-emulated
operator standing in for the HIP one (no GPU here)."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nv, n_gl, n_sph, out_dir, spectral):
    for p in (ROOT, os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd"), os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import bfsm
    import emu_lib as E
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = bfsm.reference_constants()
    f_h = bfsm.perturbed_input(bfsm.bkw_solution(nv)[0])
    gl = O.gauss_legendre(n_gl, 0.0, c["R"])
    sph = O.spherical_design(n_sph)
    shard = bfsm.shard_range(n_gl * n_sph, rank, world)
    cls = E.EmuOperatorFused if spectral == "fused" else E.EmuOperator
    op = cls(nv, gl, sph, c["gamma"], c["b_gamma"], c["L"], dir_range=shard, max_chunk=5)
    if spectral == "fused":
        spectral = False        # default route, through collidePartial
    f = torch.from_numpy(f_h.reshape(-1).copy())
    Q = torch.empty_like(f)
    if spectral == "overlap":
        # bench.py's N>1 loop: two result buffers, the all-reduce of evaluation i left in flight while evaluation i+1
        # is queued; every buffer is waited for before it is reused / read
        Qs, pending = (Q, torch.empty_like(f)), [None, None]
        for i in range(3):
            k = i & 1
            if pending[k] is not None:
                pending[k].wait()
            pending[k] = bfsm.sharded_step(op, op.qhat, Qs[k], f, dist, async_op=True)
            assert pending[k] is not None
        for w in pending:
            w.wait()
        assert torch.equal(Qs[0], Qs[1])
    else:
        assert bfsm.sharded_step(op, op.qhat, Q, f, dist, reduce_spectral=spectral) is None
    # every rank must hold the same, complete answer
    ref = O.collide(f_h, gl, sph, c["gamma"], c["b_gamma"], c["L"])
    err = float(np.abs(Q.numpy().reshape(ref.shape) - ref).max() / np.abs(ref).max())
    gathered = [torch.empty_like(Q) for _ in range(world)]
    dist.all_gather(gathered, Q)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([err, float(same), shard[0], shard[1]]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,spectral", [(2, False), (3, False), (2, True), (2, "overlap"), (2, "fused")])
def test_sharded_step_over_gloo(tmp_path, world, spectral):
    import torch.multiprocessing as mp
    nv, n_gl, n_sph = 16, 3, 12
    port = _free_port()
    mp.spawn(_worker, args=(world, port, nv, n_gl, n_sph, str(tmp_path), spectral), nprocs=world, join=True)
    covered = []
    for r in range(world):
        err, same, b0, b1 = np.load(tmp_path / f"r{r}.npy")
        assert err <= 1e-12, (r, err)
        assert same == 1.0
        covered.append((int(b0), int(b1)))
    assert covered[0][0] == 0 and covered[-1][1] == n_gl * n_sph
    assert all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))


def test_shard_range_is_balanced_and_contiguous():
    import bfsm
    for B, P in [(768, 8), (2496, 8), (5760, 8), (768, 3), (7, 8), (1, 4)]:
        parts = [bfsm.shard_range(B, r, P) for r in range(P)]
        assert parts[0][0] == 0 and parts[-1][1] == B
        assert all(parts[i][1] == parts[i + 1][0] for i in range(P - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1
