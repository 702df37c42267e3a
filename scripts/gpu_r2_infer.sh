#!/bin/bash
mkdir -p gpurun_out/r2
for cfg in "1" "0" "1" "0"; do
SEGMI_FUSE_EVAL_TOP=$cfg timeout -k 10 300 python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('fuse_eval_top=$cfg: %.2f vol/s  %.1f ms' % (d['value'], d['ms_per_step']), (d.get('roofline') or {}).get('avg_launch_ms'))"
done
