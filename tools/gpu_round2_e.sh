#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|Error|padded" $O/pytest.log | tail -15
python tools/ragged_bench.py 200 16 > $O/ragged_bench_16.log 2>&1; tail -12 $O/ragged_bench_16.log
DD_IGNORE_ABI=1 python tools/ab_builds.py decompdiff_amd/lib/libdecompdiff_hip_abi4.so decompdiff_amd/lib/libdecompdiff_hip.so 3 > $O/ab_masks.log 2>&1; cat $O/ab_masks.log
