#!/usr/bin/env python3
"""bench.py -- collision-operator evaluations per second on N MI355X GPUs (BASELINE.json's metric).

A "step" is one evaluation Q = Q(f,f) of the Fourier-spectral Boltzmann collision operator on a synthetic f
(the BKW distribution of the reference drivers, maxwell_bkw_cuda.cu:81-107) that is resident in HBM before the timed
region starts, through the C-ABI of libbfsm_hip.so -- exactly what `collision_operator(Q, f_bkw)` times in the
reference driver (maxwell_bkw_cuda.cu:144-151).

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, see self_launch())
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (the driver's form: RANK / WORLD_SIZE already in the env)

N > 1: the B = M_gl*M_sph quadrature directions are sharded contiguously over the ranks (strong scaling: the
workload is fixed); each rank computes its partial Q_gain_hat, inverse-transforms it (the transform is linear; rank 0
also subtracts the loss term) and ONE RCCL all-reduce (torch.distributed "nccl") sums the real Q over xGMI
(G doubles: half the bytes of summing Q_hat, and no kernel runs after the collective).

What `value` times, for every N: K evaluations issued in order on one stream, evaluation i+1 starting after
evaluation i -- including its collective -- has finished on the device (what a time stepper that needs Q_i to form
f_{i+1} sees; no host round trip between evaluations).  Reported beside it, never as `value`:
  blocking_call : every evaluation followed by a host synchronisation, the reference driver's timing loop
                  (maxwell_bkw_cuda.cu:144-151; bfsm_collide at N = 1);
  overlapped    : N > 1 only -- the all-reduce of evaluation i left in flight under the gain kernels of i+1.

Workloads (BASELINE.json configs):  cfg2 N=32,M_gl=8,ss009.048 | cfg3 N=64,M_gl=16,ss009.048 (default, the
roofline configuration) | cfg4 N=64,M_gl=16,ss017.156 | cfg5 N=128,M_gl=30,ss019.192 fp32.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC; RCCL between processes needs this before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd"))

WORKLOADS = {
    "cfg1": dict(nv=16, n_gl=8, n_sph=32, precision=64, design="ss007.032"),
    "cfg2": dict(nv=32, n_gl=8, n_sph=48, precision=64, design="ss009.048"),
    "cfg3": dict(nv=64, n_gl=16, n_sph=48, precision=64, design="ss009.048"),
    "cfg4": dict(nv=64, n_gl=16, n_sph=156, precision=64, design="ss017.156"),
    "cfg5": dict(nv=128, n_gl=30, n_sph=192, precision=32, design="ss019.192"),
}
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBPS = 6290.0   # measured copy ceiling recorded in the same guide (report both fractions)


def usable_cpus():
    """CPUs this process can actually run on at once: the affinity mask, capped by the cgroup CPU quota (a GPU box
    shows all 256 hardware threads of the host but grants one GPU's share of them: cpu.max = 16 CPUs)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(w, seconds_target=15.0):
    """The oracle (CPU restatement of the reference's FFTW path, kind "port") timed on this box's host cores on a
    bounded sample: the first n_sample directions of the same workload, one OpenMP thread per usable CPU."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    import bfsm
    c = bfsm.reference_constants()
    f_h = bfsm.bkw_solution(w["nv"])[0]
    gl = O.gauss_legendre(w["n_gl"], 0.0, c["R"])
    sph = O.spherical_design(w["n_sph"])
    B = w["n_gl"] * w["n_sph"]
    threads = max(1, min(O.lib().bfsm_oracle_threads(), usable_cpus()))
    n = min(B, max(2 * threads, 8))
    t0 = time.perf_counter()
    O.collide(f_h, gl, sph, c["gamma"], c["b_gamma"], c["L"], dir_range=(0, n), threads=threads)
    t1 = time.perf_counter() - t0
    # scale the sample up once if it was very short, to get ~seconds_target of CPU work
    if t1 < seconds_target / 4 and n < B:
        n2 = int(min(B, max(n, n * (seconds_target / max(t1, 1e-3)) * 0.8)))
        n2 = max(threads, (n2 // threads) * threads)
        t0 = time.perf_counter()
        O.collide(f_h, gl, sph, c["gamma"], c["b_gamma"], c["L"], dir_range=(0, n2), threads=threads)
        t1, n = time.perf_counter() - t0, n2
    evals_per_s = (n / B) / t1
    # one thread, a handful of directions (fixed part measured separately with an empty shard and subtracted)
    n1 = min(B, 8)
    t0 = time.perf_counter()
    O.collide(f_h, gl, sph, c["gamma"], c["b_gamma"], c["L"], dir_range=(0, n1), threads=1)
    t_one = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.collide(f_h, gl, sph, c["gamma"], c["b_gamma"], c["L"], dir_range=(n1, n1), threads=1)     # empty shard: fixed part
    t_fixed = time.perf_counter() - t0
    per_dir_1 = max(t_one - t_fixed, 1e-9) / n1
    return {"value": evals_per_s, "unit": "evals/s", "cores": threads, "kind": "port",
            "single_thread_value": 1.0 / (per_dir_1 * B + t_fixed),
            "sample": f"oracle/bfsm_oracle.c (own radix-2 FFT, fp64), first {n} of {B} directions of the same workload "
                      f"in {t1:.2f} s on {threads} OpenMP threads (= the CPUs this process may use: affinity mask capped "
                      f"by the cgroup quota), extrapolated linearly to B; FFTW3 is not installed in this image",
            # for context only (other hardware, the real FFTW path): the reference's own archived run
            "reference_published": "0.576 ms per direction at N=64, B=2048 on a 128-core node "
                                   "(Results/maxwell_bkw_fftw_atomics.txt:695), i.e. 0.44 s per cfg3-sized evaluation"}


def kernel_source_digest():
    """sha256 over the kernel sources: ties a PMC traffic profile to the build it was collected on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILDREN (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1) before this process has imported torch or touched the GPU, relay rank 0's JSON
    line (the children inherit stdout) and exit with the launcher's code.  Nothing is exec'ed."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--max-chunk", type=int, default=0)
    ap.add_argument("--precision", type=int, default=0, choices=(0, 32, 64),
                    help="override the workload's arithmetic type (0 = the workload's own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the extra opt-in exact-reduction measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the blocking_call / overlapped side measurements")
    ap.add_argument("--repeats", type=int, default=5,
                    help="extra K-step loops after the headline loop; their median / min / max go to the `repeats` key")
    ap.add_argument("--rehearsal", action="store_true",
                    help="CPU plumbing rehearsal of the N>1 launch path (no GPU, no measurement): the ranks are started, "
                         "form a gloo group, shard the directions and run the collective with the tests' host emulator "
                         "of the kernels on a tiny grid; the JSON line carries value = null and rehearsal = true")
    args = ap.parse_args()

    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count must equal --gpus",
                  file=sys.stderr)
        sys.exit(2)
    if args.rehearsal:
        return rehearsal(args, world, rank)

    import torch
    import bfsm

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the collision operator has no CPU path", file=sys.stderr)
        sys.exit(3)
    # one rank per GPU.  (Rehearsals on a single-GPU box may set BFSM_BENCH_BACKEND=gloo: the ranks then share
    # device 0 and the collective goes through the host -- same code path, meaningless timings.)
    backend = os.environ.get("BFSM_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        if rank == 0:
            print(f"bench.py: --gpus {world} but only {ndev} device(s) visible (BFSM_BENCH_BACKEND=gloo rehearses the "
                  "N>1 path on fewer devices; its timings mean nothing)", file=sys.stderr)
        sys.exit(4)
    dev = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # the group the measurement runs on must be the one that was asked for: never fall back to fewer ranks
        if dist.get_world_size() != args.gpus or dist.get_rank() != rank:
            print(f"bench.py: process group reports {dist.get_world_size()} rank(s) (this one {dist.get_rank()}), "
                  f"--gpus {args.gpus} / RANK {rank} were asked for", file=sys.stderr)
            sys.exit(6)

    w = WORKLOADS[args.workload]
    nv, n_gl, n_sph, prec = w["nv"], w["n_gl"], w["n_sph"], (args.precision or w["precision"])
    B = n_gl * n_sph
    c = bfsm.reference_constants()
    f_h = bfsm.bkw_solution(nv)[0]

    def make(profile, exact=False):
        op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph),
                                       nv, nv, nv, c["gamma"], c["b_gamma"], c["L"])
        op.setPrecision(prec)
        op.setDevice(dev)
        if world > 1:
            op.setDirectionShard(*bfsm.shard_range(B, rank, world))
        if args.max_chunk:
            op.setMaxChunk(args.max_chunk)
        op.setProfiling(profile)
        op.setExactReductions(exact, hermitian=exact)
        op.initialize()
        return op

    # every rank's shard and device, gathered once (reported in config; the shards must tile the B directions)
    my_shard = bfsm.shard_range(B, rank, world) if world > 1 else (0, B)
    rank_info = [{"rank": rank, "device": dev, "directions": my_shard[1] - my_shard[0]}]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rank_info[0])
        rank_info = sorted(gathered, key=lambda r: r["rank"])
        if len(rank_info) != world or sum(r["directions"] for r in rank_info) != B:
            print(f"bench.py: the ranks' shards do not tile the {B} directions: {rank_info}", file=sys.stderr)
            sys.exit(6)
        if backend == "nccl" and len({r["device"] for r in rank_info}) != world:
            print(f"bench.py: two ranks share a device: {rank_info}", file=sys.stderr)
            sys.exit(6)

    f = torch.from_numpy(f_h).cuda()
    Q = torch.empty_like(f)
    Q2 = torch.empty_like(f) if world > 1 else None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(op, mode="inorder"):
        """W warm-up + K timed evaluations, barrier + synchronize on both sides, max over ranks.
        mode "inorder"  : evaluations queued on one stream, each (with its collective) after the previous one;
             "blocking" : a host synchronisation after every evaluation (the reference driver's loop);
             "overlap"  : N>1, two result buffers, the all-reduce of evaluation i in flight under evaluation i+1."""
        qhat = bfsm.device_view(torch, *op.qhatBuffer()) if world > 1 else None
        Qs = (Q, Q2)
        pending = [None, None]

        def step(i):
            s = torch.cuda.current_stream().cuda_stream
            if world == 1:
                if mode == "blocking":
                    op.computeCollision(Q, f)            # bfsm_collide: returns when the device work has completed
                else:
                    op.computeCollisionAsync(Q, f, s)
            elif mode == "overlap":
                k = i & 1
                if pending[k] is not None:
                    pending[k].wait()
                pending[k] = bfsm.sharded_step(op, qhat, Qs[k], f, dist, s, async_op=True)
            else:
                bfsm.sharded_step(op, qhat, Q, f, dist, s)   # gain_partial -> finish_partial -> ONE RCCL all-reduce
                if mode == "blocking":
                    torch.cuda.synchronize()

        def drain():
            for k in (0, 1):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None

        for i in range(args.warmup):
            step(i)
        drain()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        drain()
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    op = make(False)
    # Set-up, not measurement: let the device reach its steady clock / page state before the W warm-up steps, so that
    # a short --steps run reports the same rate as a long one (a cold 5-step run read 6 % low).
    # Queued in bursts, like the timed loop: an evaluation that starts after an idle gap (a host synchronisation) runs its
    # first heavy kernel ~15 % slower, which would also colour a rocprofv3 --stats average of this command.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.3:
        for _ in range(8):
            op.computeCollisionAsync(Q, f, torch.cuda.current_stream().cuda_stream) if world == 1 else \
                op.gainPartial(f, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    if world > 1:
        # also set-up: RCCL builds its communicator / channels on the first collective of each kind; do that here so
        # that a --warmup 0 run does not time it
        for _ in range(2):
            dist.all_reduce(Q2)
            dist.all_reduce(Q2, async_op=True).wait()
        fence()
    elapsed = timed(op, "inorder")

    ms_per_step = 1e3 * elapsed / args.steps
    evals_per_s = args.steps / elapsed
    # `value` is the driver's K steps above.  One K-step sample of a few tens of ms moves with the chip's clock / power
    # state; the spread of further identical K-step loops is reported beside it (never as `value`).
    repeats = None
    if args.repeats > 0:
        import statistics
        rates = sorted(args.steps / timed(op, "inorder") for _ in range(args.repeats))
        repeats = {"n": args.repeats, "steps_each": args.steps, "median": statistics.median(rates), "min": rates[0],
                   "max": rates[-1], "unit": "evals/s",
                   "note": "further K-step loops after the headline loop, same handle, same timing brackets"}
    cbytes = 16.0 if prec == 64 else 8.0
    alg_bytes = (6.0 * B + 9.0) * nv ** 3 * cbytes            # whole evaluation, all GPUs
    alg_gbps = alg_bytes / (elapsed / args.steps) / 1e9

    blocking = overlapped = None
    if not args.no_extras:
        el = timed(op, "blocking")
        blocking = {"value": args.steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / args.steps,
                    "note": "host synchronisation after every evaluation (maxwell_bkw_cuda.cu:144-151)"}
        if world > 1:
            el = timed(op, "overlap")
            overlapped = {"value": args.steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / args.steps,
                          "note": "all-reduce of evaluation i in flight under the gain kernels of evaluation i+1"}

    roofline = None
    if not args.no_roofline:
        # dominant kernel, timed live with HIP events on the stream it is launched on (BFSM_FLAG_PROFILE)
        opp = make(True)
        s = torch.cuda.current_stream().cuda_stream
        def profiled_eval():          # the same entry point the timed loop uses, so the same kernels are timed
            if world == 1:
                opp.computeCollisionAsync(Q, f, s)
            else:
                opp.collidePartial(Q, f, rank == 0, s)
        for _ in range(2):
            profiled_eval()
        torch.cuda.synchronize()
        # Same regime as the timed loop: evaluations queued back to back, no host synchronisation between them; the
        # handle keeps the events of the LAST evaluation of a burst, which ran directly behind its predecessors (a
        # kernel that starts after an idle gap reads up to 15 % longer, so sum(per_kernel) would exceed ms_per_step).
        reps, acc, burst, burst_ms = 5, None, 3, 0.0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(reps):
            ev0.record()
            for _ in range(burst):
                profiled_eval()
            ev1.record()
            torch.cuda.synchronize()
            burst_ms += ev0.elapsed_time(ev1) / burst      # one PROFILED evaluation (its event records included)
            cn = opp.counters()
            cur = [(cn.kernel_ms[i], cn.kernel_alg_bytes[i], cn.kernel_launches[i]) for i in range(len(bfsm.KERNEL_NAMES))]
            acc = cur if acc is None else [(a[0] + b[0], a[1] + b[1], a[2] + b[2]) for a, b in zip(acc, cur)]
        opp.destroy()
        dom = max(range(len(acc)), key=lambda i: acc[i][0])
        ms, nbytes, launches = acc[dom]
        achieved = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
        # same command, FETCH doubled per MI355X_MICROARCH.md; profiles/summarize.py).  Valid only for the launch
        # geometry AND the kernel build it was collected on: the profile records workload, precision and a digest of
        # the kernel sources; anything else reports null (and says why) instead of a stale number.
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and world == 1 and not args.max_chunk:
            try:
                tj = json.load(open(tpath))
                same_cfg = tj.get("_workload", "cfg3") == args.workload and tj.get("_precision", 64) == prec
                if same_cfg and tj.get("_kernel_src") == kernel_source_digest():
                    traffic = tj.get(bfsm.KERNEL_NAMES[dom], {}).get("hbm_bytes_per_launch")
                    traffic_source = "profiles/" + str(tj.get("_source"))
                elif same_cfg:
                    traffic_source = f"profiles/{tj.get('_source')} is older than the kernel sources: not reported"
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": bfsm.KERNEL_NAMES[dom], "achieved": achieved, "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                    "alg_bytes_per_launch": nbytes / max(launches, 1), "avg_launch_ms": ms / max(launches, 1),
                    "launches_per_eval": launches // reps,
                    "per_kernel_regime": f"HIP events of the last of {burst} evaluations queued back to back, mean of {reps} bursts",
                    "per_kernel_sum_ms": sum(a[0] for a in acc) / reps,
                    # a profiled evaluation carries two event records per launch, so it is slightly longer than ms_per_step;
                    # the per-kernel times add up to no more than THIS figure
                    "profiled_eval_ms": burst_ms / reps,
                    "per_kernel": {bfsm.KERNEL_NAMES[i]: {"ms_per_eval": acc[i][0] / reps,
                                                          "alg_GBps": (acc[i][1] / (acc[i][0] * 1e-3) / 1e9) if acc[i][0] > 0 else 0.0}
                                   for i in range(len(acc))}}

    # Opt-in exact work reductions: same Q to rounding, ~1/3 of the FFT work.  Reported beside the
    # headline, never as the headline: `value` above always evaluates every direction -- its own two inverse 3-D
    # transforms, product and x part of the forward transform, every array written and read once per direction (6 array
    # passes); only the (y,z) part of the forward transform acts on a segment's weighted sum (config.kc_sum_before_transform).
    exact = None
    if not args.no_exact:
        ope = make(False, exact=True)
        el = timed(ope, "inorder")
        cn = ope.counters()
        exact = {"value": args.steps / el, "unit": "evals/s", "ms_per_step": 1e3 * el / args.steps,
                 "moved_bytes_per_eval_model": cn.moved_bytes_per_eval * world if world == 1 else None,
                 "antipodal_pairs_merged": bool(cn.antipodal_merged),
                 "speedup_vs_headline": (args.steps / el) / evals_per_s,
                 "note": "BFSM_FLAG_EXACT_REDUCTIONS | BFSM_FLAG_HERMITIAN: antipodal directions merged (exact for the "
                         "shipped symmetric designs), one forward FFT per radial-node segment (FFT linearity), and "
                         "only the lx >= 0 planes of A1', A2' computed/stored (f real); parity-tested to 1e-12"}
        ope.destroy()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w)

    op.destroy()
    if rank == 0:
        collective = None
        if world > 1:
            collective = {"backend": dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""),
                          "ranks": dist.get_world_size(), "per_evaluation": "1 all-reduce (sum) of Q: "
                          f"{nv ** 3 * 8} bytes per rank", "devices_visible": ndev}
        out = {
            "metric": "collision-operator evals/sec", "value": evals_per_s, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if prec == 64 else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: N={nv}^3 grid, M_gl={n_gl}, {w['design']} ({n_sph} pts), "
                                   f"B={B} directions, BKW f (t=6.5), Maxwell molecules",
                       "directions_per_gpu": B // world,
                       "directions_per_gpu_min": min(r["directions"] for r in rank_info),
                       "directions_per_gpu_max": max(r["directions"] for r in rank_info),
                       "rank_devices": [r["device"] for r in rank_info], "parallelism": f"direction-shard x{world} + 1 all-reduce",
                       "timing": "in-order: evaluation i+1 starts after evaluation i and its collective have finished",
                       "collective_overlap": False, "collective": collective,
                       # every direction: its own two inverse transforms and the x part of its forward transform; the
                       # (y,z) part of the forward transform is applied to the weighted sum of a segment's directions
                       # (linearity: same result, same bytes streamed)
                       "kc_sum_before_transform": True},
            "achieved_alg_GBps": alg_gbps, "frac_of_hbm_peak": alg_gbps / (HBM_PEAK_GBPS * world),
            "frac_of_measured_copy_ceiling": alg_gbps / (HBM_COPY_CEILING_GBPS * world),
            "alg_bytes_per_eval": alg_bytes,
            "roofline": roofline, "cpu_baseline": cpu, "blocking_call": blocking, "overlapped": overlapped,
            "exact_reductions": exact, "repeats": repeats,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def rehearsal(args, world, rank):
    """--rehearsal: the N>1 launch / shard / collective plumbing on CPU (gloo), the tests' host emulator of the kernel
    bodies standing in for the device.  Not a measurement: value and ms_per_step are null."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import torch.distributed as dist
    import bfsm
    import emu_lib as E
    from bfsm import quadrature as Qd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    nv, n_gl, n_sph = 16, 2, 12
    c = bfsm.reference_constants()
    f_h = bfsm.bkw_solution(nv)[0]
    glq, spq = bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph)
    gl = (np.asarray(glq.getNodes()), np.asarray(glq.getWeights()))
    sph = (np.asarray(spq.getx()), np.asarray(spq.gety()), np.asarray(spq.getz()), np.asarray(spq.getWeights()))
    shard = bfsm.shard_range(n_gl * n_sph, rank, world)
    op = E.EmuOperatorFused(nv, gl, sph, c["gamma"], c["b_gamma"], c["L"], dir_range=shard)
    f = torch.from_numpy(f_h.reshape(-1).copy())
    Q = torch.empty_like(f)
    for _ in range(args.warmup + args.steps):
        bfsm.sharded_step(op, op.qhat, Q, f, dist if world > 1 else None)
    whole = E.EmuOperatorFused(nv, gl, sph, c["gamma"], c["b_gamma"], c["L"])
    Qw = torch.empty_like(f)
    bfsm.sharded_step(whole, whole.qhat, Qw, f, None)
    err = float((Q - Qw).abs().max() / Qw.abs().max())
    if rank == 0:
        print(json.dumps({"metric": "collision-operator evals/sec", "value": None, "unit": "evals/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f64", "rehearsal": True,
                          "data": "REHEARSAL on CPU: host emulator of the kernels, gloo; not a measurement",
                          "config": {"workload": f"rehearsal: N={nv}, M_gl={n_gl}, {n_sph}-point design",
                                     "collective": {"backend": "gloo", "ranks": dist.get_world_size() if world > 1 else 1}},
                          "sharded_vs_whole_rel_err": err}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not err <= 1e-13:
        sys.exit(5)


if __name__ == "__main__":
    main()
