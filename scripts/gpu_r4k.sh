#!/bin/bash
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "bn_bwd_sums or mode_4 or MODE or fin" > gpurun_out/r4/k_tests.log 2>&1 || { tail -30 gpurun_out/r4/k_tests.log; exit 1; }
tail -2 gpurun_out/r4/k_tests.log
for rep in 1 2; do for v in 0 1; do
  r=$(SEGMI_BSUM32=$v timeout -k 10 300 python bench.py --workload train --size 160 --classes 32 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1) || exit 1
  echo "BSUM32=$v rep $rep $r"
done; done
