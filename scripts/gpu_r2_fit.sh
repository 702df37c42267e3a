#!/bin/bash
# fit-loop figure with and without the batch prefetch stream (alternating, one box)
mkdir -p gpurun_out/r2
for a in "10 3" "30 5"; do set -- $a
for v in 1 0 1 0; do
SEGMI_PREFETCH=$v timeout -k 10 200 python3 bench.py --workload fit --no-cpu-baseline --steps $1 --warmup $2 > gpurun_out/r2/fit.log 2>&1 || { tail -5 gpurun_out/r2/fit.log; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/r2/fit.log').read().strip().splitlines()[-1]); print('steps $1 warmup $2 prefetch $v ms_per_step', d['ms_per_step'])"
done; done
