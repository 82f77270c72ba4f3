#!/bin/bash
# round 6 (EXPERIMENTS.md R6-10; needs profiles/round6_pb_split.patch applied + a rebuild): the next layer's bond projections in two pieces on the side stream (DD_PB_SPLIT=1): A/B by environment,
# timeline of both.   usage: bash tools/gpu_round6_pbsplit.sh OUTDIR
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
L=$PWD/decompdiff_amd/lib/libdecompdiff_hip.so
python tools/ab_env.py 4 off=$L,DD_PB_SPLIT=0 split=$L,DD_PB_SPLIT=1 2>&1 | tee $O/ab.txt
DD_B=16 AB_CHECKSUM=0 python tools/ab_env.py 3 off=$L,DD_PB_SPLIT=0 split=$L,DD_PB_SPLIT=1 2>&1 | tee $O/ab_b16.txt
DD_B=1 AB_CHECKSUM=0 python tools/ab_env.py 3 off=$L,DD_PB_SPLIT=0 split=$L,DD_PB_SPLIT=1 2>&1 | tee $O/ab_b1.txt
DD_B=4 AB_CHECKSUM=0 python tools/ab_env.py 3 off=$L,DD_PB_SPLIT=0 split=$L,DD_PB_SPLIT=1 2>&1 | tee $O/ab_b4.txt
DD_WORKLOAD=large AB_CHECKSUM=0 python tools/ab_env.py 2 off=$L,DD_PB_SPLIT=0 split=$L,DD_PB_SPLIT=1 2>&1 | tee $O/ab_large.txt
for v in 0 1; do
  cd /tmp; rm -rf /tmp/tl$v
  DD_PB_SPLIT=$v rocprofv3 --kernel-trace -d /tmp/tl$v -- python $GRAFT_REPO_ROOT/tools/run_steps.py 30 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/tl$v -name "*.db" | head -1); [ -n "$f" ] && python tools/timeline.py $f 20 70 > $O/timeline_split$v.txt
  sed -n 5,22p $O/timeline_split$v.txt | cut -c1-110
done
