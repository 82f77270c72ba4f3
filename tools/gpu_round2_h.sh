#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
python -X faulthandler -m pytest tests -m gpu -q -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|Fatal|Error" $O/pytest.log | tail -8
grep -n "loss \|parameter gradients\|training losses" $O/pytest.log
