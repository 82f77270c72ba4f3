#!/bin/bash
# round 6 (EXPERIMENTS.md R6-11): hardware queues per process (GPU_MAX_HW_QUEUES, ROCm default 4) and the step graph's two branches.   usage: bash tools/gpu_round6_hwq.sh OUTDIR
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
line() { grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'value', d['value'], 'ms/step', d.get('ms_per_step'), 'steady', d.get('steady_ms_per_step'), d.get('steady_steps'), 'overhead', d.get('per_call_overhead_ms'))"; }
for q in 4 2 3 1; do
  export GPU_MAX_HW_QUEUES=$q
  for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2>/dev/null | line "hwq$q driver-style"; done
  python bench.py --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines 2>/dev/null | line "hwq$q 1000 steps"
  python bench.py --config 1 --batch 1 --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines 2>/dev/null | line "hwq$q B=1"
  python bench.py --config 3 --steps 100 --warmup 3 --no-cpu-baseline --no-rooflines 2>/dev/null | line "hwq$q cfg3"
  python bench.py --config 4 --steps 200 --warmup 10 --no-cpu-baseline --no-rooflines 2>/dev/null | line "hwq$q cfg4"
done 2>&1 | tee $O/hwq.txt
