#!/bin/bash
# round-2 evidence at the final HEAD (launch schedule 4): GPU tests, smoke, bench JSONs of every BASELINE config, kernel
# traces, one step's timeline, instruction-cache counters of the node launch
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2ev3; mkdir -p $O
python -X faulthandler -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|Fatal|Error" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_1000.json 2>> $O/bench.err
python bench.py --config 2 --steps 1000 --warmup 20 --cpu-steps 6 --cpu-warmup 1 > $O/bench_cfg2_drift.json 2>> $O/bench.err
python bench.py --config 1 --batch 16 --steps 500 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_b16.json 2>> $O/bench.err
python bench.py --config 3 --steps 100 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_100pockets.json 2>> $O/bench.err
python bench.py --config 4 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_large.json 2>> $O/bench.err
python bench.py --gpus 2 --backend gloo --config 4 --num-samples 16 --steps 50 --warmup 5 --no-cpu-baseline --no-rooflines > $O/bench_cfg4_2ranks_gloo_1gpu.json 2>> $O/bench.err
python tools/ragged_bench.py 200 16 > $O/ragged_bench_16.log 2>&1
cd /tmp
for w in small:"--steps 200 --warmup 20" large:"--workload large --steps 60 --warmup 10" b16:"--batch 16 --steps 100 --warmup 10" drift:"--config 2 --steps 200 --warmup 20"; do
  n=${w%%:*}; a=${w#*:}
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -- python $GRAFT_REPO_ROOT/bench.py $a --no-cpu-baseline --no-rooflines > $GRAFT_REPO_ROOT/$O/prof_$n.log 2>&1
done
rocprofv3 --kernel-trace -d /tmp/tl -- python $GRAFT_REPO_ROOT/tools/run_steps.py 30 > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d /tmp/pmc_ic -- python $GRAFT_REPO_ROOT/tools/run_steps.py 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for d in small large b16 drift; do f=$(find /tmp/prof_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "(round 2 final, $d)" > $O/kernel_trace_$d.md; done
f=$(find /tmp/tl -name "*.db" | head -1); [ -n "$f" ] && python tools/timeline.py $f 20 60 > $O/timeline_small.txt
f=$(find /tmp/pmc_ic -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py "$f" 6 | grep -E "^#|^\| kernel|^\|---|k_attn2_node|k_attn2_pos" > $O/pmc_icache.md
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('frac'), r.get('launch_ms'), (d.get('roofline_gemm') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
"
du -sh $O
