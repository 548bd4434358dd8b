#!/usr/bin/env python3
"""Regenerates the small Q fixtures of tests/golden/ from the parity oracle (oracle/bfsm_oracle.c).

Provenance: these vectors are produced by the CPU restatement, NOT by the reference (which cannot be built in this
image: FFTW3 / GSL / CUDA are absent).  The restatement itself is pinned by the reference's published BKW norms
(bkw_norms.json).  The fixtures freeze its output so that a later edit of the oracle or of the kernels that changes
results shows up as a diff against committed data, and so that the GPU tests also have a data-only reference.

  q_cfg1_bkw.npy     Q for BASELINE config 1 (N=16, M_gl=8, 32-point design), BKW f at t = 6.5
  q_cfg1_random.npy  same operator, seeded perturbed f (oracle.perturbed_input, seed 0x5EED, amp 0.1)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import oracle as O  # noqa: E402


def main():
    f, _, L, _ = O.bkw(16)
    gl = O.gauss_legendre(8, 0.0, 10.0)
    sph = O.spherical_design(32)
    args = (gl, sph, 0.0, 1.0 / (4.0 * np.pi), L)
    np.save(os.path.join(HERE, "q_cfg1_bkw.npy"), O.collide(f, *args, threads=1))
    np.save(os.path.join(HERE, "q_cfg1_random.npy"), O.collide(O.perturbed_input(f), *args, threads=1))
    print("written", [n for n in os.listdir(HERE) if n.endswith(".npy")])


if __name__ == "__main__":
    main()
