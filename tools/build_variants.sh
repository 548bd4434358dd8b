#!/bin/bash
# Builds A/B variants of the product library for kernel experiments:  tools/build_variants.sh name:"-DFLAG ..." ...
# Output: gpurun_variants/libbfsm_<name>.so (travels to the GPU box; select with BFSM_LIB=...).  Not part of the product.
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/boltzmann-fourier-spectral-method_amd
mkdir -p $R/gpurun_variants
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -fno-slp-vectorize $f -shared -o $R/gpurun_variants/libbfsm_$n.so $P/csrc/bfsm_hip.hip &
done
wait
ls -la $R/gpurun_variants
