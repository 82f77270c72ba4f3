#!/bin/bash
# round 4, E2: v-side tables stored channel-permuted (v pass: 2 x 16-byte reads per member row and lane instead of 8 x 4-byte)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_e2; mkdir -p $O
L=decompdiff_amd/lib
V="base=$L/libdecompdiff_hip.so,DD_VPERM=0 vperm=$L/libdecompdiff_hip_vperm.so,DD_VPERM=1"
python tools/ab_env.py 3 $V 2>&1 | tee $O/ab.txt
AB_CHECKSUM=0 DD_B=16 python tools/ab_env.py 2 $V 2>&1 | tee $O/ab_b16.txt
AB_CHECKSUM=0 DD_WORKLOAD=large python tools/ab_env.py 2 $V 2>&1 | tee $O/ab_large.txt
AB_CHECKSUM=0 DD_WORKLOAD=mid python tools/ab_env.py 2 $V 2>&1 | tee $O/ab_mid.txt
