#!/bin/bash
mkdir -p gpurun_out/r4
for rep in 1 2; do
for v in 0 21 2122 11 1122; do
  SEGMI_WGRAD_CT22=$v timeout -k 10 300 python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r4/ct22_${v}_$rep.json 2>/dev/null || exit 1
  echo "CT22=$v rep $rep: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4/ct22_${v}_$rep.json | head -1)"
done
done
