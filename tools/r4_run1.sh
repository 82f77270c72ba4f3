#!/bin/bash
# round 4, call 1: E3 (bond-layer tail trips spread, default) vs -DDD_BL_TAIL=0, then the GPU test suite at this tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_run1; mkdir -p $O
L=decompdiff_amd/lib
bash tools/ab_libs.sh $O $L/libdecompdiff_hip_notail.so $L/libdecompdiff_hip.so
DD_B=16 python tools/ab_builds.py $L/libdecompdiff_hip_notail.so $L/libdecompdiff_hip.so 2 2>&1 | tee $O/ab_b16.txt
DD_WORKLOAD=large python tools/ab_builds.py $L/libdecompdiff_hip_notail.so $L/libdecompdiff_hip.so 2 2>&1 | tee $O/ab_large.txt
DD_WORKLOAD=mid python tools/ab_builds.py $L/libdecompdiff_hip_notail.so $L/libdecompdiff_hip.so 2 2>&1 | tee $O/ab_mid.txt
DD_B=4 python tools/ab_builds.py $L/libdecompdiff_hip_notail.so $L/libdecompdiff_hip.so 2 2>&1 | tee $O/ab_b4.txt
python -X faulthandler -m pytest tests -m gpu -q -x -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|Fatal|Error" $O/pytest.log | tail -8
