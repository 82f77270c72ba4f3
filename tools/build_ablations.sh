#!/bin/bash
# Timing-only builds of the attention kernels with parts removed: tools/ablate_node.patch (applied to a scratch copy of
# csrc/) adds -DDD_ABLATE=<mask> switches -- 1 table MFMAs (+ the features feeding them), 2 score MFMAs, 4 aggregation
# MFMAs, 8 row gathers, 16 LayerNorms, 32 query fold, 64 epilogue mat-vec, 128 in-kernel query MLP of the coordinate launch, 256 its v-pass gathers, 512 weight-image staging.  Results are wrong for mask != 0.
#   tools/build_ablations.sh 0 1 2 4 8 16 32 64 127   ->  decompdiff_amd/lib/libdd_abl_<mask>.so
#   DD_HIP_LIB=$PWD/decompdiff_amd/lib/libdd_abl_1.so python tools/ablate_node.py      (on an MI355X)
cd "$(dirname "$0")/.." || exit 1
python -m decompdiff_amd.build > /dev/null || exit 1
L=$PWD/decompdiff_amd/lib; F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on"
T=$(mktemp -d); mkdir -p $T/decompdiff_amd $T/include; cp -r decompdiff_amd/csrc $T/decompdiff_amd/; cp include/*.h $T/include/
( cd $T && patch -p1 -s < "$OLDPWD/tools/ablate_node.patch" ) || exit 1
for m in "$@"; do
  ( /opt/rocm/bin/hipcc $F -DDD_ABLATE=$m -c $T/decompdiff_amd/csrc/dd_attention2.hip -o $L/dd_attention2_abl$m.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/dd_gemm.o $L/dd_graph.o $L/dd_attention2_abl$m.o $L/dd_step.o $L/dd_scatter.o $L/dd_api.o -o $L/libdd_abl_$m.so && echo built $m ) &
done
wait; rm -rf $T
