#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "prefetch or cli_train or augmentation" > gpurun_out/r2/t_fit.log 2>&1; tail -4 gpurun_out/r2/t_fit.log
for v in 1 0 1 0; do
SEGMI_PREFETCH=$v timeout -k 10 200 python3 bench.py --workload fit --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r2/fit.log 2>&1 || { tail -5 gpurun_out/r2/fit.log; exit 1; }
python3 -c "import json; d=json.loads(open('gpurun_out/r2/fit.log').read().strip().splitlines()[-1]); print('prefetch $v ms_per_step', d['ms_per_step'])"
done
