import os, sys, time
sys.path.insert(0, "boltzmann-fourier-spectral-method_amd"); sys.path.insert(0, ".")
import torch, bfsm
from bench import WORKLOADS
c = bfsm.reference_constants()
for name, P in (("cfg1", 1), ("cfg2", 1), ("cfg3", 8), ("cfg3", 1)):
    w = WORKLOADS[name]; nv, n_gl, n_sph = w["nv"], w["n_gl"], w["n_sph"]
    f = torch.from_numpy(bfsm.bkw_solution(nv)[0]).cuda(); Q = torch.empty_like(f); Qg = torch.empty_like(f)
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0.0, c["R"]), bfsm.SphericalDesign(n_sph), nv, nv, nv, c["gamma"], c["b_gamma"], c["L"])
    op.setDirectionShard(*bfsm.shard_range(n_gl * n_sph, 0, P)); op.initialize()
    def step(Qo, s):
        op.gainPartial(f, s); op.finishPartial(Qo, f, True, s)
    s0 = torch.cuda.current_stream().cuda_stream
    for _ in range(20): step(Q, s0)
    torch.cuda.synchronize()
    n = 2000 if nv < 64 else 300
    t0 = time.perf_counter()
    for _ in range(n): step(Q, s0)
    torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / n
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step(Qg, side.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        step(Qg, torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    print(name, P, "graph result equal:", torch.equal(Q, Qg))
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / n
    print(f"{name} P={P}: plain {t_plain*1e6:.1f} us/eval, graph {t_graph*1e6:.1f} us/eval", flush=True)
    op.destroy()
