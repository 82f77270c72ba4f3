#!/bin/bash
# Builds of the library that differ in the compiler flags of dd_attention2.hip only (scheduling options: results stay
# bit-identical).  usage: bash tools/flag_sweep_build.sh   -> decompdiff_amd/lib/var_<name>.so
cd "$(dirname "$0")/.." || exit 1
L=decompdiff_amd/lib; C=decompdiff_amd/csrc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on"
declare -A V=(
 [maxilp]="-mllvm -amdgpu-sched-strategy=max-ilp"
 [maxilp_nohighrp]="-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1"
 [maxilp_trackers]="-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-use-amdgpu-trackers=1"
 [maxilp_nolowocc]="-mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-disable-clustered-low-occupancy-reschedule=1"
 [iterilp]="-mllvm -amdgpu-sched-strategy=iterative-ilp"
 [itermaxocc]="-mllvm -amdgpu-sched-strategy=iterative-maxocc"
 [nohighrp]="-mllvm -amdgpu-disable-unclustered-high-rp-reschedule=1"
)
for n in "${!V[@]}"; do
  ( /opt/rocm/bin/hipcc $BASE ${V[$n]} -c $C/dd_attention2.hip -o /tmp/var_$n.o 2> /tmp/var_$n.err &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/dd_gemm.o $L/dd_graph.o /tmp/var_$n.o $L/dd_step.o $L/dd_scatter.o $L/dd_api.o -o $L/var_$n.so &&
    echo "built $n" || { echo "FAILED $n"; tail -3 /tmp/var_$n.err; } ) &
done
wait
ls -la $L/var_*.so
