#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t4.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t4.log
tail -5 gpurun_out/r3/t4.log
grep -q "pytest rc=0" gpurun_out/r3/t4.log || exit 1
for i in 1 2 3; do
  SEGMI_FUSE_FIN=0 timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse_fin=0', round(d['ms_per_step'],3))"
  SEGMI_FUSE_FIN=1 timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fuse_fin=1', round(d['ms_per_step'],3))"
done
echo "--- phase times (default)"; timeout -k 10 120 python scripts/phase_times.py 20 2>/dev/null
echo "--- phase times (no weight gradients issued: main chain alone; WRONG gradients, timing only)"; SEGMI_DIAG_SKIP_WGRAD=1 timeout -k 10 120 python scripts/phase_times.py 20 2>/dev/null
echo "--- phase times serial"; SEGMI_SERIAL=1 timeout -k 10 120 python scripts/phase_times.py 20 2>/dev/null
