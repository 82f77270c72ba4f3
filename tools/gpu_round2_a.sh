#!/bin/bash
# round-2 GPU session A: full GPU test suite, driver-style bench, exact-math study, bench of the other configs, traces
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-steps 4 --cpu-warmup 1 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_1000.json 2> $O/bench_1000.err
python bench.py --config 2 --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg2.json 2>> $O/bench.err
python bench.py --config 1 --batch 16 --steps 500 --warmup 20 --no-cpu-baseline > $O/bench_b16.json 2>> $O/bench.err
python bench.py --config 3 --pockets 8 --steps 200 --warmup 5 --no-cpu-baseline > $O/bench_cfg3_8pockets.json 2>> $O/bench.err
python bench.py --config 4 --num-samples 16 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_16samples.json 2>> $O/bench.err
python bench.py --workload large --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_large_b8.json 2>> $O/bench.err
# exact-math variant: the 1000-step chains with correctly rounded rsqrt / softmax division
DD_HIP_LIB=$PWD/decompdiff_amd/lib/libdecompdiff_hip_exact.so python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "1000_steps" > $O/exact_1000.log 2>&1
DD_HIP_LIB=$PWD/decompdiff_amd/lib/libdecompdiff_hip_exact.so python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-rooflines > $O/bench_exact.json 2>> $O/bench.err
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-rooflines > $O/bench_default_300.json 2>> $O/bench.err
# kernel traces
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_small -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-rooflines > $GRAFT_REPO_ROOT/$O/prof_small.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_large -- python $GRAFT_REPO_ROOT/bench.py --workload large --steps 60 --warmup 10 --no-cpu-baseline --no-rooflines > $GRAFT_REPO_ROOT/$O/prof_large.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_small prof_large; do f=$(find $O/$d -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "($d)" > $O/$d.md; done
find $O -name "*.db" -size +8M -delete
du -sh $O
