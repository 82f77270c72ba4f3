#!/bin/bash
# round 6: parity subset, then A/B of {VALU-only build, + lin_node in the node launch, + four-wave coordinate launch}, GEMM tile clocks,
# kernel trace + timeline of the default build.   usage: bash tools/gpu_round6_ab.sh OUTDIR
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -30 > $O/parity.txt; tail -4 $O/parity.txt
L=decompdiff_amd/lib
python tools/ab_builds.py $L/libdecompdiff_hip_nocoop.so $L/libdecompdiff_hip_noquad.so $L/libdecompdiff_hip_oldgemm.so $L/libdecompdiff_hip.so 3 2>&1 | tee $O/ab.txt
DD_B=1 python tools/ab_builds.py $L/libdecompdiff_hip_nocoop.so $L/libdecompdiff_hip_noquad.so $L/libdecompdiff_hip_oldgemm.so $L/libdecompdiff_hip.so 2 2>&1 | tee $O/ab_b1.txt
DD_B=16 python tools/ab_builds.py $L/libdecompdiff_hip_nocoop.so $L/libdecompdiff_hip_noquad.so $L/libdecompdiff_hip_oldgemm.so $L/libdecompdiff_hip.so 2 2>&1 | tee $O/ab_b16.txt
python tools/gemm_clocks.py > $O/gemm_clocks.txt 2>&1; cat $O/gemm_clocks.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_small -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-rooflines --no-steady > $GRAFT_REPO_ROOT/$O/prof_small.log 2>&1
rocprofv3 --kernel-trace -d /tmp/tl -- python $GRAFT_REPO_ROOT/tools/run_steps.py 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tl -name "*.db" | head -1); [ -n "$f" ] && python tools/timeline.py $f 20 70 > $O/timeline_small.txt
f=$(find /tmp/prof_small -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "(round 6, $1, small)" > $O/kernel_trace_small.md
head -14 $O/kernel_trace_small.md; sed -n 1,30p $O/timeline_small.txt
