#!/bin/bash
# final check at HEAD: full GPU suite, smoke, the driver's bench command, the drift / B=16 lines with calibrated launch timing
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2final; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
grep -E "passed|failed|FAILED|Fatal|Error" $O/pytest.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err
python bench.py --config 2 --steps 1000 --warmup 20 --cpu-steps 6 --cpu-warmup 1 > $O/bench_cfg2_drift.json 2>> $O/bench.err
python bench.py --config 1 --batch 16 --steps 500 --warmup 20 --no-cpu-baseline > $O/bench_cfg1_b16.json 2>> $O/bench.err
python -c "
import json
for f in ['bench_driver_style','bench_cfg2_drift','bench_cfg1_b16']:
    d=json.loads([l for l in open('$O/'+f+'.json') if l.startswith('{')][-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['frac'], r['launch_ms'], d['roofline_gemm']['frac'], (d.get('cpu_baseline') or {}).get('value'))
"
