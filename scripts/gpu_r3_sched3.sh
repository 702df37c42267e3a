#!/bin/bash
mkdir -p gpurun_out/r3
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2; do
  run "default (ws 128, mfma 128)    " X=1
  run "mfma grids for  64 CUs        " SEGMI_WGRAD_CUS_MFMA=64
  run "mfma grids for  96 CUs        " SEGMI_WGRAD_CUS_MFMA=96
  run "mfma grids for 192 CUs        " SEGMI_WGRAD_CUS_MFMA=192
  run "mfma grids for 256 CUs        " SEGMI_WGRAD_CUS_MFMA=256
  run "ws 160, mfma 128              " SEGMI_WGRAD_CUS=160 SEGMI_WGRAD_CUS_MFMA=128
  run "ws 96, mfma 128               " SEGMI_WGRAD_CUS=96 SEGMI_WGRAD_CUS_MFMA=128
done 2>&1 | tee gpurun_out/r3/sched3_ab.txt
