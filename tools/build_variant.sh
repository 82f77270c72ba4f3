#!/bin/bash
# usage: tools/build_variant.sh NAME "<extra hipcc flags for dd_attention2.hip>"  ->  decompdiff_amd/lib/libdecompdiff_hip_NAME.so
# (the other objects come from the default build: run `python -m decompdiff_amd.build` first)
set -e
cd "$(dirname "$0")/.."
L=decompdiff_amd/lib; C=decompdiff_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $2 -c $C/dd_attention2.hip -o $L/dd_attention2_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/dd_gemm.o $L/dd_graph.o $L/dd_attention2_$1.o $L/dd_step.o $L/dd_scatter.o $L/dd_train.o $L/dd_api.o -o $L/libdecompdiff_hip_$1.so
echo $L/libdecompdiff_hip_$1.so
