#!/bin/bash
mkdir -p gpurun_out/r4
bash scripts/gpu_check.sh || exit 1
for rep in 1 2; do
for v in 128 96 160 192; do
  SEGMI_WGRAD_CUS=$v timeout -k 10 300 python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r4/wcus_${v}_$rep.json 2>/dev/null || exit 1
  echo "WGRAD_CUS=$v rep $rep: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4/wcus_${v}_$rep.json | head -1)"
done
done
