#!/bin/bash
# round 6 (EXPERIMENTS.md R6-11): priority of the library's side stream (DD_SIDE_PRIO: 1 = lowest (rounds 3-6), 0 = default priority) on every bench line.   usage: bash tools/gpu_round6_side_prio.sh OUTDIR
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
line() { grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'value', d['value'], 'ms/step', d.get('ms_per_step'), 'steady', d.get('steady_ms_per_step'), d.get('steady_steps'), 'overhead', d.get('per_call_overhead_ms'))"; }
for rep in 1 2; do for p in 1 0; do
  export DD_SIDE_PRIO=$p
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2>/dev/null | line "prio$p driver-style"
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2>/dev/null | line "prio$p driver-style"
done; done 2>&1 | tee $O/prio.txt
for p in 1 0; do
  export DD_SIDE_PRIO=$p
  python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines 2>/dev/null | line "prio$p 1000 steps"
  python bench.py --config 2 --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines 2>/dev/null | line "prio$p drift"
  python bench.py --config 1 --batch 1 --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines 2>/dev/null | line "prio$p B=1"
  python bench.py --config 1 --batch 16 --steps 500 --warmup 20 --no-cpu-baseline --no-rooflines 2>/dev/null | line "prio$p B=16"
  python bench.py --config 3 --steps 100 --warmup 3 --no-cpu-baseline --no-rooflines 2>/dev/null | line "prio$p cfg3"
  python bench.py --config 4 --steps 200 --warmup 10 --no-cpu-baseline --no-rooflines 2>/dev/null | line "prio$p cfg4"
done 2>&1 | tee -a $O/prio.txt
