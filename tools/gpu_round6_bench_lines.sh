#!/bin/bash
# round 6: the driver's bench line three times + the long runs with the final bench.py (steady figure from the SECOND long call).   usage: bash tools/gpu_round6_bench_lines.sh OUTDIR
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_gpu_configs.py -q -k "bench_json_line_contract" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
for i in 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines > $O/bench_driver_style_$i.json 2>> $O/bench.err; done
python bench.py --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_1000.json 2>> $O/bench.err
python bench.py --config 2 --steps 1000 --warmup 20 --no-cpu-baseline > $O/bench_cfg2_drift.json 2>> $O/bench.err
python bench.py --config 1 --batch 16 --steps 500 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_b16.json 2>> $O/bench.err
python bench.py --config 1 --batch 1 --steps 1000 --warmup 20 --no-cpu-baseline --no-rooflines > $O/bench_cfg1_b1.json 2>> $O/bench.err
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
    print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('steady_ms_per_step'), d.get('steady_steps'), d.get('per_call_overhead_ms'), r.get('frac'), r.get('launch_ms'), r.get('traffic'))
"
