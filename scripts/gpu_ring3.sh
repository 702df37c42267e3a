#!/bin/bash
# ring3 bring-up: conv / unet / e2e parity tests, then ring2 vs ring3 A/B of the training step and inference
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv or ring or bn_bwd or sums or affine" > gpurun_out/r4/ring3_ops.log 2>&1; rc=$?
tail -15 gpurun_out/r4/ring3_ops.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_unet_gpu.py tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r4/ring3_e2e.log 2>&1; rc=$?
tail -15 gpurun_out/r4/ring3_e2e.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
  for v in 0 1; do
    SEGMI_RING3=$v timeout -k 10 300 python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r4/ring3_ab_train_${v}_$i.json 2>/dev/null || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/r4/ring3_ab_train_${v}_$i.json").read().strip().splitlines()[-1])
print("RING3=$v run $i train ms", round(d["ms_per_step"],3), "top conv ms", round(d["roofline"]["avg_launch_ms"],4), d["roofline"]["kernel"][:40])
PY
  done
done
