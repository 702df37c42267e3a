#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t5.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t5.log
tail -4 gpurun_out/r3/t5.log
grep -q "pytest rc=0" gpurun_out/r3/t5.log || exit 1
timeout -k 10 200 python scripts/fin_tail_bench.py 2>/dev/null | tee gpurun_out/r3/fin_tail_bench.txt
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2; do
  run "default          " X=1
  run "wgrad_cus=192     " SEGMI_WGRAD_CUS=192
  run "mask192 + cus192  " SEGMI_SIDE_CUS=192 SEGMI_WGRAD_CUS=192
  run "mask224 + cus224  " SEGMI_SIDE_CUS=224 SEGMI_WGRAD_CUS=224
  run "mask128 + cus128  " SEGMI_SIDE_CUS=128 SEGMI_WGRAD_CUS=128
  run "mask192, no defer " SEGMI_SIDE_CUS=192 SEGMI_WGRAD_CUS=192 SEGMI_DEFER_TOP_WGRAD=0
  run "no defer          " SEGMI_DEFER_TOP_WGRAD=0
done 2>&1 | tee gpurun_out/r3/cumask_ab.txt
