#!/bin/bash
# A/B of the driver-style line (bench.py --steps 20 --warmup 5): allocator settings / result-tensor capacity.  usage: tools/first_call_ab.sh [reps]
R=${1:-4}
line() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in $(seq $R); do
  echo -n "default            : "; line
  echo -n "final at num_steps : "; DD_FINAL_CAP=0 line
  echo -n "heap-only malloc   : "; MALLOC_MMAP_THRESHOLD_=536870912 MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=67108864 line
  echo -n "heap-only, no cap  : "; DD_FINAL_CAP=0 MALLOC_MMAP_THRESHOLD_=536870912 MALLOC_TRIM_THRESHOLD_=1073741824 MALLOC_TOP_PAD_=67108864 line
done
