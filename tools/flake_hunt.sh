#!/bin/bash
# repeat the GPU config tests (fresh process each time) and report crashes: hunting a rare host-side segfault inside hipEventSynchronize
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/flake; mkdir -p $O
N=${1:-8}
for i in $(seq 1 $N); do
  python -X faulthandler -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "not rccl and not eight_ranks and not two_ranks and not bench_json" > $O/run_$i.log 2>&1
  echo "run $i rc=$?" | tee -a $O/summary.txt
done
grep -l "Fatal Python error" $O/run_*.log | head
