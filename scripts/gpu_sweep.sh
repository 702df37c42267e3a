#!/bin/bash
# alternating runs of the training step under single-variable settings: gpu_sweep.sh "VAR=val" "VAR=val" ... (first = baseline "")
mkdir -p gpurun_out/r4
out=gpurun_out/r4/sweep_$(date +%H%M%S).txt
for rep in 1 2; do
  for cfg in "$@"; do
    r=$(env $cfg timeout -k 10 300 python bench.py --workload ${WL:-train} --steps 30 --warmup 5 --no-cpu-baseline --no-lane-ab 2>/dev/null | grep -o '"ms_per_[a-z]*": [0-9.]*' | head -1) || exit 1
    echo "rep $rep [$cfg] $r" | tee -a $out
  done
done
