#!/bin/bash
# round 3 closing run: C4-shaped single-GPU step, the profile set, then the whole suite once more
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python bench.py --workload train --size 160 --classes 32 --batch 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 shape B=2 160^3 K=32:', round(d['ms_per_step'],2), 'ms', round(d['value']/1e9,3), 'Gvox/s', d['roofline']['kernel'][:80], round(d['roofline']['frac'],3))" | tee gpurun_out/r3/c4_single.txt
bash scripts/gpu_profiles.sh 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t9.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t9.log
tail -3 gpurun_out/r3/t9.log
