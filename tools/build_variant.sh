#!/bin/bash
# usage: [SRC=dd_gemm] tools/build_variant.sh NAME "<extra hipcc flags for the one source>"  ->  decompdiff_amd/lib/libdecompdiff_hip_NAME.so
# A second library that differs in the compile flags of ONE source (default dd_attention2.hip; SRC=dd_gemm: dd_gemm.hip); the other
# objects come from the default build: run `python -m decompdiff_amd.build` first.
set -e
cd "$(dirname "$0")/.."
L=decompdiff_amd/lib; C=decompdiff_amd/csrc; S=${SRC:-dd_attention2}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $2 -c $C/$S.hip -o $L/${S}_$1.o
OBJS=""
for o in dd_gemm dd_graph dd_attention2 dd_step dd_scatter dd_train dd_api; do
  if [ "$o" = "$S" ]; then OBJS="$OBJS $L/${S}_$1.o"; else OBJS="$OBJS $L/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $L/libdecompdiff_hip_$1.so
echo $L/libdecompdiff_hip_$1.so
