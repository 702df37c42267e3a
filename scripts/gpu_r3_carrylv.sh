#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -k "schedules or golden or convergence" > gpurun_out/r3/t21.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t21.log
tail -3 gpurun_out/r3/t21.log
grep -q "pytest rc=0" gpurun_out/r3/t21.log || exit 1
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
  run "carry 1 level " SEGMI_CARRY_LEVELS=1
  run "carry 2 levels" SEGMI_CARRY_LEVELS=2
  run "carry 3 levels" SEGMI_CARRY_LEVELS=3
  run "carry 4 levels" SEGMI_CARRY_LEVELS=4
done 2>&1 | tee gpurun_out/r3/carrylv_ab.txt
for e in "SEGMI_CARRY_LEVELS=1" "SEGMI_CARRY_LEVELS=2" "SEGMI_CARRY_LEVELS=3"; do echo "== $e"; env $e timeout -k 10 150 python scripts/phase_times.py 20 2>&1 | grep -v amdgpu | tail -6; done | tee gpurun_out/r3/phase_times2.txt
