#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t3.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t3.log
tail -5 gpurun_out/r3/t3.log
grep -q "pytest rc=0" gpurun_out/r3/t3.log || exit 1
for v in train_cpu train_sleep train; do
  timeout -k 10 200 python scripts/probe_cpu_then_infer.py $v 2>/dev/null | tail -1 | tee -a gpurun_out/r3/probe_cpu2.jsonl
done
for i in 1 2; do
  SEGMI_FUSE_FIN=0 timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse_fin=0', round(d['ms_per_step'],3))"
  SEGMI_FUSE_FIN=1 timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse_fin=1', round(d['ms_per_step'],3))"
done
