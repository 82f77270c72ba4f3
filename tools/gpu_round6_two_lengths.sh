#!/bin/bash
# round 6 (EXPERIMENTS.md R6-11): the second chain length of a process runs 14 % slower -- which knob moves it.   usage: bash tools/gpu_round6_two_lengths.sh
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r6two; mkdir -p $O
run() { echo "## $1 | $2"; env $1 python tools/two_lengths.py $2 2>&1 | grep steps | sed 's/ steps: /:/; s/ ms\/step//' | tr '\n' ';'; echo; }
for p in 0 1; do
  run "DD_SIDE_PRIO=$p" "100 200 8"
  run "DD_SIDE_PRIO=$p" "500 1000 8"
  run "DD_SIDE_PRIO=$p" "200 400 16"
  run "DD_SIDE_PRIO=$p" "200 400 1"
  run "DD_SIDE_PRIO=$p DD_CHAIN_CACHE=0" "200 400 8"
  run "DD_SIDE_PRIO=$p GPU_MAX_HW_QUEUES=8" "200 400 8"
  run "DD_SIDE_PRIO=$p GPU_MAX_HW_QUEUES=5" "200 400 8"
done | tee $O/knobs2.txt
