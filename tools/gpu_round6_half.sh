#!/bin/bash
# round 6 (EXPERIMENTS.md R6-9b; needs profiles/round6_ne_half_blocks.patch applied + a rebuild incl. the trace variant): trailing node blocks of the fused launch as two half blocks (DD_NE_HALF=n), NB blocks first (DD_NB_FIRST=1): A/B by environment,
# occupancy traces.   usage: bash tools/gpu_round6_half.sh OUTDIR
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
L=$PWD/decompdiff_amd/lib/libdecompdiff_hip.so
python tools/ab_env.py 1 base=$L h48nbf=$L,DD_NE_HALF=48,DD_NB_FIRST=1 2>&1 | tee $O/checksums.txt
AB_CHECKSUM=0 python tools/ab_env.py 3 base=$L h48=$L,DD_NE_HALF=48 nbf=$L,DD_NB_FIRST=1 h16nbf=$L,DD_NE_HALF=16,DD_NB_FIRST=1 h32nbf=$L,DD_NE_HALF=32,DD_NB_FIRST=1 h48nbf=$L,DD_NE_HALF=48,DD_NB_FIRST=1 h64nbf=$L,DD_NE_HALF=64,DD_NB_FIRST=1 2>&1 | tee $O/ab.txt
DD_B=16 AB_CHECKSUM=0 python tools/ab_env.py 2 base=$L h48=$L,DD_NE_HALF=48 nbf=$L,DD_NB_FIRST=1 h32nbf=$L,DD_NE_HALF=32,DD_NB_FIRST=1 h64nbf=$L,DD_NE_HALF=64,DD_NB_FIRST=1 2>&1 | tee $O/ab_b16.txt
T=$PWD/decompdiff_amd/lib/libdecompdiff_hip_trace.so
for v in "0 0" "48 0" "0 1" "32 1" "48 1"; do
  set -- $v
  echo "## DD_NE_HALF=$1 DD_NB_FIRST=$2" | tee -a $O/node_trace.txt
  DD_NE_HALF=$1 DD_NB_FIRST=$2 DD_HIP_LIB=$T python tools/node_trace.py 8 2>&1 | grep -v amdgpu.ids | tee -a $O/node_trace.txt | cut -c1-250
done
cat ~/.cache/decompdiff_amd/*.txt 2>/dev/null | sort | uniq -c | sort -rn | head -30
