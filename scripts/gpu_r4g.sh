#!/bin/bash
mkdir -p gpurun_out/r4
bash scripts/gpu_ab_step.sh SEGMI_WGRAD_CT32 0 21 train 2 || exit 1
bash scripts/gpu_ab_step.sh SEGMI_WGRAD_CT22 0 21 train 2 || exit 1
for v in 0 21; do
  SEGMI_WGRAD_CT22=$v timeout -k 10 300 python bench.py --workload train --size 160 --classes 32 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4/c4_ct22_$v.json 2>gpurun_out/r4/c4_ct22_$v.err || { tail -5 gpurun_out/r4/c4_ct22_$v.err; exit 1; }
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4/c4_ct22_$v.json | head -1
done
