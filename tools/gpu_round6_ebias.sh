#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
# round 6 (EXPERIMENTS.md R6-5c): GEMM tile with only the bias requested ahead of the MFMAs; variant built with SRC=dd_gemm tools/build_variant.sh ebias -DDD_GEMM_EARLY_BIAS=1 against a default of 0
export TMPDIR=/tmp
O=gpurun_out/r6ebias; mkdir -p $O
L=decompdiff_amd/lib
python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_ebias.so 4 2>&1 | tee $O/ab.txt
DD_B=1 python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_ebias.so 3 2>&1 | tee $O/ab_b1.txt
DD_B=16 python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_ebias.so 3 2>&1 | tee $O/ab_b16.txt
