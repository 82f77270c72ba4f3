#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "ragged_groups_together" > $O/t1.log 2>&1; echo "alone rc=$?"
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "ragged" > $O/t2.log 2>&1; echo "ragged tests rc=$?"
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -s -x -k "cache or padded or ragged" > $O/t3.log 2>&1; echo "cache+padded+ragged rc=$?"
DD_CHAIN_CACHE=0 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -s -x -k "cache or padded or ragged" > $O/t4.log 2>&1; echo "no cache rc=$?"
tail -3 $O/t1.log $O/t2.log $O/t3.log $O/t4.log | cut -c1-200
