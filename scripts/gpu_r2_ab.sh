#!/bin/bash
# same-box A/B of the training step: $1 = env var to toggle (0 / 1), e.g. SEGMI_WGRAD_WS
mkdir -p gpurun_out/r2
v=$1
for rep in 1 2; do for x in 0 1; do
  env $v=$x timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2> gpurun_out/r2/ab.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v=$x ms/step %.3f  top fwd %.1f us' % (d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3))"
done; done
