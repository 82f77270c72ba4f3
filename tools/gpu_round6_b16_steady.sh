#!/bin/bash
# round 6: why a long call behind a job of another length is slow (second live graph of the process).   usage: bash tools/gpu_round6_b16_steady.sh
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --config 1 --batch ${B:-16} --steps $1 --warmup 20 --no-cpu-baseline --no-rooflines 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$2 B', d['config']['batch_per_unit'], 'steps', d['steps'], 'ms/step', d['ms_per_step'], 'steady', d['steady_ms_per_step'], d['steady_steps'])"; }
run 500 default
GPU_MAX_HW_QUEUES=8 run 500 hwq8
GPU_MAX_HW_QUEUES=2 run 500 hwq2
DD_CHAIN_CACHE=0 run 500 nocache
B=8 run 500 default
B=8 GPU_MAX_HW_QUEUES=8 run 500 hwq8
