#!/bin/bash
# round-5 evidence at HEAD: GPU tests (+ the full chains once more on the -DDD_EXACT_MATH build), smoke, bench JSONs of every
# BASELINE config, kernel traces, PMC passes of the node launch, one step's timeline.  usage: bash tools/gpu_round5_evidence.sh [notests]
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5fin; mkdir -p $O
sha256sum decompdiff_amd/csrc/dd_attention2.hip | cut -c1-16 > $O/kernel_source_sha256_16.txt
if [ "$1" != "notests" ]; then
  python -X faulthandler -m pytest tests -m gpu -q -s --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
  grep -E "passed|failed|FAILED|Fatal|Error" $O/pytest.log | tail -5
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
  # the two B = 8 full chains (and the single-sample ones) on the exact-math build: does v_rsq_f32 / the shared reciprocal move
  # the step at which samples leave 1e-4?  (records land in gpurun_out/tests/parity_full_chain.json under "exact_math")
  if [ -f decompdiff_amd/lib/libdecompdiff_hip_exact.so ]; then
    DD_HIP_LIB=$PWD/decompdiff_amd/lib/libdecompdiff_hip_exact.so python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -s \
      -k "full_chain_at_the_bench_shape or trajectory_1000_steps_golden" > $O/pytest_exact_math.log 2>&1
    grep -E "passed|failed" $O/pytest_exact_math.log | tail -2
  fi
fi
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
for i in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines > $O/bench_driver_style_$i.json 2>> $O/bench.err; done
DD_BENCH_TRACE=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2> $O/bench_call_trace.txt > /dev/null
python tools/phase_clocks.py > $O/phase_clocks.txt 2>&1
tools/_build/graph_churn_repro 200 6 > $O/graph_churn_repro.txt 2>&1
python bench.py --gpus 8 --config 3 --plan-only > $O/plan_8ranks_cfg3.json 2>> $O/bench.err
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_1000.json 2>> $O/bench.err
python bench.py --config 2 --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_drift.json 2>> $O/bench.err
python bench.py --config 1 --batch 16 --steps 500 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_b16.json 2>> $O/bench.err
python bench.py --config 3 --steps 100 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_100pockets.json 2>> $O/bench.err
python bench.py --config 4 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_large.json 2>> $O/bench.err
cd /tmp
for w in small:"--steps 200 --warmup 20" large:"--workload large --steps 60 --warmup 10"; do
  n=${w%%:*}; a=${w#*:}
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$n -- python $GRAFT_REPO_ROOT/bench.py $a --no-cpu-baseline --no-rooflines > $GRAFT_REPO_ROOT/$O/prof_$n.log 2>&1
done
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for pmc in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_small_$i -- python $GRAFT_REPO_ROOT/tools/run_steps.py 12 > $GRAFT_REPO_ROOT/$O/pmc_small_$i.log 2>&1
done
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  DD_WORKLOAD=large rocprofv3 --pmc $pmc --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_large_$i -- python $GRAFT_REPO_ROOT/tools/run_steps.py 6 > $GRAFT_REPO_ROOT/$O/pmc_large_$i.log 2>&1
done
rocprofv3 --kernel-trace -d /tmp/tl -- python $GRAFT_REPO_ROOT/tools/run_steps.py 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tl -name "*.db" | head -1); [ -n "$f" ] && python tools/timeline.py $f 20 60 > $O/timeline_small.txt
for d in small large; do f=$(find $O/prof_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "(round 5, HEAD, $d)" > $O/kernel_trace_$d.md; done
for d in $O/pmc_*; do [ -d "$d" ] || continue; f=$(find $d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py "$f" 6 > $d.md; done
find $O -name "*.db" -delete; find $O -type d -empty -delete
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d['ms_per_step'], r.get('frac'), r.get('launch_ms'), (d.get('roofline_gemm') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
"
du -sh $O
