#!/bin/bash
# sliding-window inference with 1 / 2 / 3 window-group lanes + the sliding-window tests
python -m pytest tests -x -q -m gpu > gpurun_out/t_lanes.log 2>&1; tail -2 gpurun_out/t_lanes.log
for n in 2; do
  echo "lanes=$n"
  SEGMI_SW_LANES=$n python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
done
