#!/bin/bash
# round 4, E4: one fork per layer (dd_debug_set_option(27, 1): the side stream forms the new h itself) vs the shipped schedule 4
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_e4; mkdir -p $O
python tools/ab_bench.py "27=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_b8.txt
DD_B=16 python tools/ab_bench.py "27=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_b16.txt
DD_B=1 python tools/ab_bench.py "27=1" 2>&1 | grep -v amdgpu.ids | tee $O/ab_b1.txt
