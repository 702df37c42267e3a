#!/bin/bash
mkdir -p gpurun_out/r2
for g in 1 2 4; do for l in 2 3; do
SEGMI_SW_GROUP=$g SEGMI_SW_LANES=$l timeout -k 10 300 python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('group x$g lanes $l: %.2f vol/s  %.1f ms  top conv %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3))"
done; done
