#!/bin/bash
# round-2 evidence, part 2: PMC passes of the node launch, the 100-pocket job, the 1000-step bench with calibrated event timing
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2ev; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench2.err
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_1000.json 2>> $O/bench2.err
python bench.py --config 3 --steps 100 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_100pockets.json 2>> $O/bench2.err
python bench.py --config 4 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_large.json 2>> $O/bench2.err
cd /tmp
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
cd $GRAFT_REPO_ROOT
for d in $O/pmc_*; do [ -d "$d" ] || continue; f=$(find $d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_pmc.py "$f" 6 > $d.md; done
find $O -name "*.db" -delete; find $O -type d -empty -delete
ls $O; tail -3 $O/bench2.err
