#!/bin/bash
# round 6, after the side stream moved to the default priority (R6-11): GPU suite, smoke, every bench line, the two-lengths check, kernel trace + timeline.
# usage: bash tools/gpu_round6_final2.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6fin2; mkdir -p $O
sha256sum decompdiff_amd/csrc/dd_attention2.hip | cut -c1-16 > $O/kernel_source_sha256_16.txt
python -X faulthandler -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|ERROR" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
for i in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines > $O/bench_driver_style_$i.json 2>> $O/bench.err; done
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_1000.json 2>> $O/bench.err
python bench.py --config 2 --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_drift.json 2>> $O/bench.err
python bench.py --config 1 --batch 16 --steps 500 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_b16.json 2>> $O/bench.err
python bench.py --config 1 --batch 1 --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines > $O/bench_cfg1_b1.json 2>> $O/bench.err
python bench.py --config 3 --steps 100 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_100pockets.json 2>> $O/bench.err
rm -rf /tmp/nscache; DD_NODE_SPLIT_CACHE_DIR=/tmp/nscache python bench.py --config 3 --steps 1000 --cold --no-cpu-baseline --no-rooflines > $O/bench_cfg3_cold_cachecold.json 2>> $O/bench.err
DD_NODE_SPLIT_CACHE_DIR=/tmp/nscache python bench.py --config 3 --steps 1000 --cold --no-cpu-baseline --no-rooflines > $O/bench_cfg3_cold_cachewarm.json 2>> $O/bench.err
python bench.py --config 4 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_large.json 2>> $O/bench.err
for b in 8 1; do for p in 0 1; do echo "## DD_SIDE_PRIO=$p B=$b"; DD_SIDE_PRIO=$p python tools/two_lengths.py 200 400 $b 2>&1 | grep steps | sed 's/ steps: /:/; s/ ms\/step//' | tr '\n' ';'; echo; done; done > $O/two_lengths.txt; cat $O/two_lengths.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_small -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-rooflines --no-steady > $GRAFT_REPO_ROOT/$O/prof_small.log 2>&1
rocprofv3 --kernel-trace -d /tmp/tl -- python $GRAFT_REPO_ROOT/tools/run_steps.py 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tl -name "*.db" | head -1); [ -n "$f" ] && python tools/timeline.py $f 20 60 > $O/timeline_small.txt
f=$(find $O/prof_small -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "(round 6, HEAD, small)" > $O/kernel_trace_small.md
find $O -name "*.db" -delete; find $O -type d -empty -delete
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('steady_ms_per_step'), d.get('per_call_overhead_ms'), r.get('frac'), r.get('launch_ms'), r.get('traffic'), d.get('per_shape_setup_ms'), d.get('setup_fraction_of_cold'))
    except Exception as e: print(f, 'ERR', e)
"
head -8 $O/kernel_trace_small.md | cut -c1-160
