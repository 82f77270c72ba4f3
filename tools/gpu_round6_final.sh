#!/bin/bash
# last pass of round 6: the GPU suite at HEAD, the FETCH_SIZE / WRITE_SIZE passes of the node launch on the FINAL kernel source (bench.py ties
# roofline.traffic to its hash), one driver-style bench line.   usage: bash tools/gpu_round6_final.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6last; mkdir -p $O
sha256sum decompdiff_amd/csrc/dd_attention2.hip | cut -c1-16 > $O/kernel_source_sha256_16.txt
python -X faulthandler -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|ERROR" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp
i=2
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_small_$i -- python $GRAFT_REPO_ROOT/tools/run_steps.py 12 > $GRAFT_REPO_ROOT/$O/pmc_small_$i.log 2>&1
done
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  DD_WORKLOAD=large rocprofv3 --pmc $pmc --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_large_$i -- python $GRAFT_REPO_ROOT/tools/run_steps.py 6 > $GRAFT_REPO_ROOT/$O/pmc_large_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
for d in $O/pmc_*; do [ -d "$d" ] || continue; f=$(find $d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py "$f" 6 > $d.md; done
find $O -name "*.db" -delete; find $O -type d -empty -delete
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
grep -h k_attn2_node $O/pmc_*.md | cut -c1-160
python -c "
import json
d=json.loads([l for l in open('$O/bench_driver_style.json') if l.startswith('{')][-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], d['steady_ms_per_step'], d['per_call_overhead_ms'], r['frac'], r['launch_ms'], r['traffic'], r['mfma_instructions_per_launch'])
"
