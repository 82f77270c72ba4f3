#!/bin/bash
# round 6: bench.py with other --steps / --warmup than the driver's (the warm-up calls size the cached trajectory buffers for the timed call: model.traj_capacity_hint)
cd "$GRAFT_REPO_ROOT"
line() { grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'value', d['value'], 'ms/step', d.get('ms_per_step'), 'steady', d.get('steady_ms_per_step'), 'overhead', d.get('per_call_overhead_ms'))"; }
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2>/dev/null | line "20/5"
python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-rooflines 2>/dev/null | line "100/10"
python bench.py --gpus 1 --steps 50 --warmup 3 --no-cpu-baseline --no-rooflines 2>/dev/null | line "50/3"
python -m pytest tests/test_gpu_configs.py -q -k "bench_json_line or chain_cache" 2>&1 | tail -2
