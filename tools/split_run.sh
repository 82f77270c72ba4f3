mkdir -p gpurun_out/r2split
for Q in default 2 3 4 8; do
  for F in 1 3; do
    if [ "$Q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$Q; fi
    echo "== GPU_MAX_HW_QUEUES=$Q fusion=$F" >> gpurun_out/r2split/log.txt
    timeout 300 python tools/split_bench.py 200 $F >> gpurun_out/r2split/log.txt 2>&1
  done
done
cat gpurun_out/r2split/log.txt
