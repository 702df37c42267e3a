#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 120 python scripts/dice_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3/dice_bench.txt
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r3/t19.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t19.log
tail -3 gpurun_out/r3/t19.log
grep -q "pytest rc=0" gpurun_out/r3/t19.log || exit 1
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', round(d['ms_per_step'],3))"
done 2>&1 | tee gpurun_out/r3/ew.txt
bash scripts/gpu_sertime.sh
