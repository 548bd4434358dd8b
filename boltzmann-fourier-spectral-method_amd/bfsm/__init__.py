"""Python plumbing over the C-ABI of libbfsm_hip.so (include/bfsm.h).

This is NOT the product: the product is the HIP library and its C++ mirror of the reference's operator class
(host/HIPBoltzmannOperator.hpp).  This package exists so that tests/ and bench.py can drive the same C-ABI entry
points from Python, with torch used only for device memory, streams and torch.distributed (RCCL).

There is deliberately no CPU fallback: if libbfsm_hip.so is missing or no GPU is visible, construction fails.
"""
from .capi import (BFSM_F32, BFSM_F64, BFSM_FLAG_EXACT_REDUCTIONS, BFSM_FLAG_PROFILE, KERNEL_NAMES, BfsmError, Counters, Desc, lib_path,
                   load_library)
from .operator import HIPBoltzmannOperator, shard_range
from .sharded import device_view, sharded_step
from .quadrature import GaussLegendreQuadrature, SphericalDesign
from .bkw import bkw_solution, error_norms, reference_constants, perturbed_input
from .relax import relax_bkw, ssp_rk3_step

__all__ = [
    "BFSM_F32", "BFSM_F64", "BFSM_FLAG_PROFILE", "BFSM_FLAG_EXACT_REDUCTIONS", "KERNEL_NAMES", "BfsmError", "Counters", "Desc", "lib_path",
    "load_library", "HIPBoltzmannOperator", "shard_range", "sharded_step", "device_view", "GaussLegendreQuadrature", "SphericalDesign",
    "bkw_solution", "error_norms", "reference_constants", "perturbed_input", "relax_bkw", "ssp_rk3_step",
]
