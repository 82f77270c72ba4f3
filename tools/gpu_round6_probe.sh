#!/bin/bash
# round-6 probe: driver-style bench line, 200-step line, phase clocks, kernel trace + one step's timeline.  usage: bash tools/gpu_round6_probe.sh OUTDIR
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines > $O/bench_cfg1_1000.json 2>> $O/bench.err
python tools/phase_clocks.py > $O/phase_clocks.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_small -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-rooflines > $GRAFT_REPO_ROOT/$O/prof_small.log 2>&1
rocprofv3 --kernel-trace -d /tmp/tl -- python $GRAFT_REPO_ROOT/tools/run_steps.py 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tl -name "*.db" | head -1); [ -n "$f" ] && python tools/timeline.py $f 20 70 > $O/timeline_small.txt
f=$(find /tmp/prof_small -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "(round 6, $1, small)" > $O/kernel_trace_small.md
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('frac'), r.get('launch_ms'), (d.get('roofline_gemm') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
"
head -12 $O/kernel_trace_small.md; cat $O/phase_clocks.txt | tail -14
