#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
python tools/short_calls.py 20 6 > $O/short_calls.log 2>&1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 20 steps:', d['value'], d['ms_per_step'])"; done > $O/bench20.log 2>&1
OMP_NUM_THREADS=8 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 20 steps OMP=8:', d['value'], d['ms_per_step'])" >> $O/bench20.log 2>&1
cat $O/short_calls.log $O/bench20.log
