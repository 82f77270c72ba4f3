#!/bin/bash
# per-unit host time of a short 100-pocket-style job (configs[3]: a new shape per unit), a few runs: shows the sporadic
# per-unit stalls (pinned allocations, graph instantiation) next to the 20 x ~2 ms of GPU work
for i in 1 2 3; do python bench.py --config 3 --pockets 16 --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' value', d['value'], ' ms/unit:', [round(1e3*u['seconds_enqueue'],1) for u in d['per_unit']])
"; done
