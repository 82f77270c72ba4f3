#!/bin/bash
# round-6 evidence at HEAD: GPU tests, smoke, bench JSONs of every BASELINE config (+ configs[3] --cold with the node-split cache cold and
# warm), kernel traces, PMC passes of the node launch, one step's timeline, phase clocks, the node launch's occupancy trace.
# usage: bash tools/gpu_round6_evidence.sh [notests]
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6fin; mkdir -p $O
sha256sum decompdiff_amd/csrc/dd_attention2.hip | cut -c1-16 > $O/kernel_source_sha256_16.txt
if [ "$1" != "notests" ]; then
  python -X faulthandler -m pytest tests -m gpu -q -s --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
  grep -E "passed|failed|FAILED|Fatal|Error" $O/pytest.log | tail -5
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
for i in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines > $O/bench_driver_style_$i.json 2>> $O/bench.err; done
python tools/phase_clocks.py > $O/phase_clocks.txt 2>&1
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_1000.json 2>> $O/bench.err
python bench.py --config 2 --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_drift.json 2>> $O/bench.err
python bench.py --config 1 --batch 16 --steps 500 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_b16.json 2>> $O/bench.err
python bench.py --config 1 --batch 1 --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines > $O/bench_cfg1_b1.json 2>> $O/bench.err
python bench.py --config 3 --steps 100 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_100pockets.json 2>> $O/bench.err
rm -rf /tmp/nscache; DD_NODE_SPLIT_CACHE_DIR=/tmp/nscache python bench.py --config 3 --steps 1000 --cold --no-cpu-baseline --no-rooflines > $O/bench_cfg3_cold_cachecold.json 2>> $O/bench.err
DD_NODE_SPLIT_CACHE_DIR=/tmp/nscache python bench.py --config 3 --steps 1000 --cold --no-cpu-baseline --no-rooflines > $O/bench_cfg3_cold_cachewarm.json 2>> $O/bench.err
python bench.py --config 4 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_large.json 2>> $O/bench.err
[ -f decompdiff_amd/lib/libdecompdiff_hip_trace.so ] && for B in 8 16; do DD_HIP_LIB=$PWD/decompdiff_amd/lib/libdecompdiff_hip_trace.so python tools/node_trace.py $B 2>&1 | grep -v amdgpu.ids; done > $O/node_trace.txt
cd /tmp
for w in small:"--steps 200 --warmup 20" large:"--workload large --steps 60 --warmup 10"; do
  n=${w%%:*}; a=${w#*:}
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$n -- python $GRAFT_REPO_ROOT/bench.py $a --no-cpu-baseline --no-rooflines --no-steady > $GRAFT_REPO_ROOT/$O/prof_$n.log 2>&1
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
for d in small large; do f=$(find $O/prof_$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "(round 6, HEAD, $d)" > $O/kernel_trace_$d.md; done
for d in $O/pmc_*; do [ -d "$d" ] || continue; f=$(find $d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py "$f" 6 > $d.md; done
find $O -name "*.db" -delete; find $O -type d -empty -delete
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('steady_ms_per_step'), d.get('per_call_overhead_ms'), r.get('frac'), r.get('launch_ms'), (d.get('roofline_gemm') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), d.get('per_shape_setup_ms'), d.get('setup_fraction_of_cold'))
    except Exception as e: print(f, 'ERR', e)
"
du -sh $O
