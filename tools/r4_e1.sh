#!/bin/bash
# round 4, E1: persistent bond-layer trips in source-major order (-DDD_BL_SRC_MAJOR=1) against the default build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_e1; mkdir -p $O
L=decompdiff_amd/lib
bash tools/ab_libs.sh $O $L/libdecompdiff_hip.so $L/libdecompdiff_hip_srcmaj.so
DD_B=16 python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_srcmaj.so 2 2>&1 | tee $O/ab_b16.txt
DD_WORKLOAD=large python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_srcmaj.so 2 2>&1 | tee $O/ab_large.txt
DD_WORKLOAD=mid python tools/ab_builds.py $L/libdecompdiff_hip.so $L/libdecompdiff_hip_srcmaj.so 2 2>&1 | tee $O/ab_mid.txt
