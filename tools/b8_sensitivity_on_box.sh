#!/bin/bash
# The oracle's +-1-ulp replays of the two bench-shape chains (traj1000_b8_plain / traj1000_b8_drift: 8 samples, 1000 steps), three
# seeds each, as six parallel CPU processes (16 threads each: ~5.6 s per oracle step -> ~95 min) on a many-core host.  Run through
# gpurun only because that box has 256 CPUs (this container has 8: 6 replays would take > 8 h); no GPU is used.
# Outputs: gpurun_out/sens/sens_<fixture>_<seed>.npz, merged by tools/merge_b8_sensitivity.py into tests/golden/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sens
for name in traj1000_b8_plain traj1000_b8_drift; do
  for seed in 9000 9001 9002; do
    python -m oracle.make_sensitivity --name $name --runs 1 --first-seed $seed --threads 16 --out gpurun_out/sens/sens_${name}_${seed}.npz \
      > gpurun_out/sens/log_${name}_${seed}.txt 2>&1 &
  done
done
wait
tail -n 2 gpurun_out/sens/log_*.txt
