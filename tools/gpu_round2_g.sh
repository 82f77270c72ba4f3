#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
python -X faulthandler -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|Fatal" $O/pytest.log | tail -5
if ! grep -q "passed" $O/pytest.log; then DD_CHAIN_CACHE=0 python -X faulthandler -m pytest tests -m gpu -q -s -x > $O/pytest_nocache.log 2>&1; echo "nocache rc=$?"; grep -E "passed|failed|FAILED|Fatal" $O/pytest_nocache.log | tail -5; fi
DD_IGNORE_ABI=1 python tools/ab_builds.py decompdiff_amd/lib/libdecompdiff_hip_abi4.so decompdiff_amd/lib/libdecompdiff_hip.so 3 > $O/ab_masks.log 2>&1; cat $O/ab_masks.log
