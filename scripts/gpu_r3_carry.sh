#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t6.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t6.log
tail -4 gpurun_out/r3/t6.log
grep -q "pytest rc=0" gpurun_out/r3/t6.log || exit 1
timeout -k 10 200 python scripts/fin_tail_bench.py 2>/dev/null | tee gpurun_out/r3/fin_tail_bench2.txt
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
  run "carry=0           " SEGMI_CARRY_TOP_WGRAD=0
  run "carry=1 (default) " X=1
  run "carry=1 defer=0   " SEGMI_DEFER_TOP_WGRAD=0
  run "carry=1 fin=0     " SEGMI_FUSE_FIN=0
done 2>&1 | tee gpurun_out/r3/carry_ab.txt
timeout -k 10 300 python bench.py --workload fit --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fit', round(d['ms_per_step'],3))"
