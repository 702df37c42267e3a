#!/bin/bash
mkdir -p gpurun_out/r3
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2; do
  run "default (defer, carry)        " X=1
  run "no defer                      " SEGMI_DEFER_TOP_WGRAD=0
  run "no carry                      " SEGMI_CARRY_TOP_WGRAD=0
  run "no defer, no carry            " SEGMI_DEFER_TOP_WGRAD=0 SEGMI_CARRY_TOP_WGRAD=0
  run "defer depth 1                 " SEGMI_DEFER_DEPTH=1
  run "defer depth 2                 " SEGMI_DEFER_DEPTH=2
  run "defer depth 3                 " SEGMI_DEFER_DEPTH=3
  run "8 hardware queues             " GPU_MAX_HW_QUEUES=8
done 2>&1 | tee gpurun_out/r3/sched_ab.txt
