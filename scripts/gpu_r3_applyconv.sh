#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "inside_the_stride2 or one_launch" > gpurun_out/r3/t12.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t12.log
tail -5 gpurun_out/r3/t12.log
grep -q "pytest rc=0" gpurun_out/r3/t12.log || exit 1
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py tests/test_unet_gpu.py -m gpu -x -q > gpurun_out/r3/t13.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t13.log
tail -5 gpurun_out/r3/t13.log
grep -q "pytest rc=0" gpurun_out/r3/t13.log || exit 1
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
  run "apply in conv=0 " SEGMI_FUSE_APPLY_CONV=0
  run "apply in conv=1 " X=1
done 2>&1 | tee gpurun_out/r3/applyconv_ab.txt
bash scripts/gpu_sertime.sh
