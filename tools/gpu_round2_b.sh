#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -15
python tools/call_overhead.py 20 4 > $O/call_overhead.log 2>&1
python tools/phase_clocks.py > $O/phase_clocks.log 2>&1
tail -30 $O/call_overhead.log
