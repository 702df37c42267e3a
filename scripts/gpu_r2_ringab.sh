#!/bin/bash
# A/B of the ring kernel's XCD-aware column map inside the real workloads (one box, alternating)
mkdir -p gpurun_out/r2
for v in 1 0 1 0; do
SEGMI_RING2_XCD=$v timeout -k 10 200 python3 bench.py --workload train --no-cpu-baseline --steps 30 --warmup 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd $v ms_per_step', d['ms_per_step'], 'roof', d['roofline']['achieved'])" || exit 1
done
for v in 1 0 1 0; do
SEGMI_RING2_XCD=$v timeout -k 10 200 python3 bench.py --workload infer --no-cpu-baseline --steps 4 --warmup 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('xcd $v infer', d['value'], d['unit'])" || exit 1
done
