#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "one_launch or bn_" > gpurun_out/r3/t10.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t10.log
tail -3 gpurun_out/r3/t10.log
grep -q "pytest rc=0" gpurun_out/r3/t10.log || exit 1
timeout -k 10 120 python scripts/fused_bn_bench.py 2>&1 | tee gpurun_out/r3/fused_bn_bench.txt
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
  run "bn small fused=0 " SEGMI_FUSE_BN_BWD_SMALL=0
  run "bn small fused=1 " X=1
done 2>&1 | tee gpurun_out/r3/bnfused2_ab.txt
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py tests/test_unet_gpu.py -m gpu -x -q > gpurun_out/r3/t11.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t11.log
tail -3 gpurun_out/r3/t11.log
