#!/bin/bash
# round 6 (EXPERIMENTS.md R6-12): LayerNorm beta' of a tile requested from LDS ahead of the variance reduction (needs profiles/round6_ln_preload.patch applied; tools/build_variant.sh lnpre -DDD_LN_PRELOAD=1).   usage: bash tools/gpu_round6_lnpre.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6lnpre; mkdir -p $O
L=decompdiff_amd/lib
python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_lnpre.so 4 2>&1 | tee $O/ab.txt
DD_B=16 python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_lnpre.so 3 2>&1 | tee $O/ab_b16.txt
DD_WORKLOAD=large python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_lnpre.so 2 2>&1 | tee $O/ab_large.txt
