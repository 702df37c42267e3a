#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t20.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t20.log
tail -3 gpurun_out/r3/t20.log
grep -q "pytest rc=0" gpurun_out/r3/t20.log || exit 1
run() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py "${BARGS[@]}" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
BARGS=(--workload train --steps 20)
for c in 256 144 136 128 120; do run "train, wgrad grids for $c CUs" SEGMI_WGRAD_CUS=$c; done 2>&1 | tee gpurun_out/r3/wcus_ab3.txt
run "train, default              " X=1 | tee -a gpurun_out/r3/wcus_ab3.txt
BARGS=(--workload fit --steps 20)
for c in 256 128; do run "fit, wgrad grids for $c CUs" SEGMI_WGRAD_CUS=$c; done 2>&1 | tee -a gpurun_out/r3/wcus_ab3.txt
BARGS=(--workload train --size 160 --classes 32 --batch 2 --steps 5 --warmup 2)
for c in 256 128; do run "C4 shape, wgrad grids for $c CUs" SEGMI_WGRAD_CUS=$c; done 2>&1 | tee -a gpurun_out/r3/wcus_ab3.txt
BARGS=(--workload train --precision f32 --steps 5 --warmup 2)
for c in 256 128; do run "f32, wgrad grids for $c CUs" SEGMI_WGRAD_CUS=$c; done 2>&1 | tee -a gpurun_out/r3/wcus_ab3.txt
