#!/bin/bash
# Archives the C++ driver's stdout in the reference's Results/ format:  bash tools/results_logs.sh TAG  (on the GPU box)
# -> gpurun_out/results_TAG/maxwell_bkw_hip_*.txt ; copy into profiles/results_TAG/.
TAG=${1:?tag}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/results_$TAG; P=$R/boltzmann-fourier-spectral-method_amd
mkdir -p $O
D="--design-dir $P/data/sph_design"
$P/maxwell_bkw_hip --Nv 32 --Ns 12 -t 20 --warmup 3 $D > $O/maxwell_bkw_hip_Nv32_Ns12.txt 2>&1
$P/maxwell_bkw_hip --Nv 32 --Ns 32 -t 20 --warmup 3 $D > $O/maxwell_bkw_hip_Nv32_Ns32.txt 2>&1
$P/maxwell_bkw_hip --Nv 48 --Ns 12 -t 20 --warmup 3 $D > $O/maxwell_bkw_hip_Nv48_Ns12.txt 2>&1
$P/maxwell_bkw_hip --Nv 64 --Ns 12 -t 20 --warmup 3 $D > $O/maxwell_bkw_hip_Nv64_Ns12.txt 2>&1
$P/maxwell_bkw_hip --Nv 64 --Ns 32 -t 20 --warmup 3 $D > $O/maxwell_bkw_hip_Nv64_Ns32.txt 2>&1
$P/maxwell_bkw_hip --Nv 96 --Ngl 16 --Ns 12 -t 5 --warmup 2 $D > $O/maxwell_bkw_hip_Nv96_Ngl16_Ns12.txt 2>&1
$P/maxwell_bkw_hip --Nv 16 --Ngl 8 --Ns 32 -t 20 --warmup 3 $D > $O/maxwell_bkw_hip_cfg1.txt 2>&1
$P/maxwell_bkw_hip --Nv 64 --Ngl 16 --Ns 48 -t 20 --warmup 3 $D > $O/maxwell_bkw_hip_cfg3.txt 2>&1
grep -H "L2 error\|evals_per_s" $O/*.txt
