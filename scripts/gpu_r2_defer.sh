#!/bin/bash
# deferred issue of the upper levels' weight gradients (single GPU; default on): A/B, alternating runs
mkdir -p gpurun_out/r2
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r2/t_all.log 2>&1; tail -2 gpurun_out/r2/t_all.log
grep -q " passed" gpurun_out/r2/t_all.log || exit 1
for i in 1 2 3; do for v in 1 0; do
SEGMI_DEFER_TOP_WGRAD=$v timeout -k 10 200 python3 bench.py --workload train --no-cpu-baseline --steps 40 --warmup 5 > gpurun_out/r2/ab.log 2>&1 || { tail -5 gpurun_out/r2/ab.log; exit 1; }
python3 -c "import sys,json; d=json.loads(open('gpurun_out/r2/ab.log').read().strip().splitlines()[-1]); print('defer $v train ms_per_step', d['ms_per_step'])"
done; done
for v in 1 0 1 0; do
SEGMI_DEFER_TOP_WGRAD=$v timeout -k 10 200 python3 bench.py --workload fit --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r2/ab.log 2>&1 || { tail -5 gpurun_out/r2/ab.log; exit 1; }
python3 -c "import sys,json; d=json.loads(open('gpurun_out/r2/ab.log').read().strip().splitlines()[-1]); print('defer $v fit ms_per_step', d['ms_per_step'])"
done
