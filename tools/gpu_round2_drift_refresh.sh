#!/bin/bash
# refresh of the drift evidence after the arm-scaffold kernel was parallelised (bench line + kernel trace)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2drift; mkdir -p $O
python bench.py --config 2 --steps 1000 --warmup 20 --cpu-steps 6 --cpu-warmup 1 > $O/bench_cfg2_drift.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_drift -- python $GRAFT_REPO_ROOT/bench.py --config 2 --steps 200 --warmup 20 --no-cpu-baseline --no-rooflines > $GRAFT_REPO_ROOT/$O/prof_drift.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_drift -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "(round 2 final, arm-scaffold drift kernel parallelised, drift)" > $O/kernel_trace_drift.md
grep -E "drift_armsca|drift_clash" $O/kernel_trace_drift.md | cut -c1-150
python -c "
import json
d=json.loads([l for l in open('$O/bench_cfg2_drift.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
