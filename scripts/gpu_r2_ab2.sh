#!/bin/bash
for v in 256 128 192 256 128; do
  SEGMI_WGRAD_WS_CUS=$v timeout -k 10 300 python bench.py --workload train --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('ws cus $v: ms/step %.3f' % d['ms_per_step'])"
done
