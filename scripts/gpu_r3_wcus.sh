#!/bin/bash
mkdir -p gpurun_out/r3
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2; do
  for c in 256 160 128 112 96 80 64 48 32; do
    run "wgrad grids for $c CUs " SEGMI_WGRAD_CUS=$c
  done
done 2>&1 | tee gpurun_out/r3/wcus_ab2.txt
