#!/bin/bash
mkdir -p gpurun_out/r2
for cfg in "4 4" "4 0" "4 1" "1 0" "4 4" "4 0" "8 4" "8 2"; do
set -- $cfg
SEGMI_SW_GROUP=$1 SEGMI_EVAL_TOP_CHUNK=$2 timeout -k 10 300 python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('group $1 top chunk $2: %.2f vol/s  %.1f ms' % (d['value'], d['ms_per_step']))"
done
