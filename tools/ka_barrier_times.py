#!/usr/bin/env python3
"""Cycles the waves of KA's pipelined-pair loop spend inside each of its 12 barriers per direction.

Needs the instrumented build:  bash tools/build_variants.sh bartimes:"-DBFSM_TOOLS_BUILD -DBFSM_KA_BARRIER_TIMES"  (+ -DBFSM_KA_XLANE: the cross-lane form, 8 barriers)
usage: BFSM_LIB=gpurun_variants/libbfsm_bartimes.so python3 tools/ka_barrier_times.py c5s [cfg3]
Barrier k of an iteration (csrc/bfsm_core.hpp, body_gain_inv, pipelined pair):
  0 before wr A (previous direction's last reads)   1 wr A done -> rd A     2 rd A done -> wr B
  3 wr B done -> rd B          4 rd B done -> wr^T A      5 wr^T A done -> rd^T A      6 rd^T A done -> wr^T B
  7 wr^T B done -> rd^T B      8 rd^T B done -> wr A      9 wr A done -> rd A         10 rd A done -> wr B
 11 wr B done (behind the stores of tile A) -> rd B
"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "boltzmann-fourier-spectral-method_amd"))
import torch
import bfsm

NAMED = {"cfg3": (64, 16, 48, 64), "cfg4": (64, 16, 156, 64), "c5s": (128, 4, 192, 32)}
c = bfsm.reference_constants()
for case in sys.argv[1:]:
    nv, n_gl, n_sph, prec = NAMED[case]
    f = torch.from_numpy(bfsm.bkw_solution(nv)[0]).cuda()
    Q = torch.empty_like(f)
    op = bfsm.HIPBoltzmannOperator(bfsm.GaussLegendreQuadrature(n_gl, 0, c["R"]), bfsm.SphericalDesign(n_sph), nv, nv, nv,
                                   c["gamma"], c["b_gamma"], c["L"])
    op.setPrecision(prec)
    op.initialize()
    L = op._lib
    L.bfsm_debug_counters.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 32)()
    for _ in range(20):
        op(Q, f)
    torch.cuda.synchronize()
    L.bfsm_debug_counters(out, 1)
    evals = 10
    for _ in range(evals):
        op(Q, f)
    torch.cuda.synchronize()
    L.bfsm_debug_counters(out, 0)
    waves, total = out[14], out[13]
    print(f"{case}: {waves // evals} waves per evaluation, {total / waves:.0f} s_memtime ticks per wave in the direction loop (100 MHz ticks? see ratio only)")
    tb = [out[k] for k in range(12)]
    print("  share of the wave's loop time inside each barrier (incl. the wait for its own outstanding LDS / memory operations):")
    print("  " + " ".join(f"b{k}={100.0 * tb[k] / total:.1f}%" for k in range(12)))
    print(f"  all barriers: {100.0 * sum(tb) / total:.1f} %")
    op.destroy()
