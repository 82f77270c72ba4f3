#!/bin/bash
# round 6 (EXPERIMENTS.md R6-9a; needs profiles/round6_bl_late.patch applied + a rebuild incl. the trace variant): persistent bond-layer workgroups behind the node blocks (DD_BL_LATE): A/B by environment, occupancy trace.   usage: bash tools/gpu_round6_late.sh OUTDIR
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
L=$PWD/decompdiff_amd/lib
python tools/ab_env.py 4 off=$L/libdecompdiff_hip.so,DD_BL_LATE=0 late=$L/libdecompdiff_hip.so,DD_BL_LATE=1 2>&1 | tee $O/ab.txt
DD_B=16 AB_CHECKSUM=0 python tools/ab_env.py 3 off=$L/libdecompdiff_hip.so,DD_BL_LATE=0 late=$L/libdecompdiff_hip.so,DD_BL_LATE=1 2>&1 | tee $O/ab_b16.txt
DD_B=4 AB_CHECKSUM=0 python tools/ab_env.py 3 off=$L/libdecompdiff_hip.so,DD_BL_LATE=0 late=$L/libdecompdiff_hip.so,DD_BL_LATE=1 2>&1 | tee $O/ab_b4.txt
DD_WORKLOAD=large AB_CHECKSUM=0 python tools/ab_env.py 2 off=$L/libdecompdiff_hip.so,DD_BL_LATE=0 late=$L/libdecompdiff_hip.so,DD_BL_LATE=1 2>&1 | tee $O/ab_large.txt
for late in 0 1; do
  DD_BL_LATE=$late DD_HIP_LIB=$L/libdecompdiff_hip_trace.so python tools/node_trace.py 8 > $O/node_trace_late$late.txt 2>&1; cat $O/node_trace_late$late.txt | cut -c1-250
done
cat ~/.cache/decompdiff_amd/*.txt 2>/dev/null | sort | uniq -c | tail -20; ls ~/.cache/decompdiff_amd/
