#!/bin/bash
mkdir -p gpurun_out/r4
for v in 0 11 21; do
  SEGMI_WGRAD_CT32=$v timeout -k 10 300 python bench.py --workload train --size 160 --classes 32 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/c4_ct32_$v.json 2>gpurun_out/r4/c4_ct32_$v.err || { tail -5 gpurun_out/r4/c4_ct32_$v.err; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4/c4_ct32_$v.json | head -1
done
SEGMI_WGRAD_CT32=11 timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -3
