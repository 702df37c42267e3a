#!/bin/bash
# sliding-window tests + inference bench (+ kernel stats of the serial run)
python -m pytest tests -x -q -m gpu -k "sliding or blend or slab or ensemble or class_count" > gpurun_out/t_lanes.log 2>&1; tail -2 gpurun_out/t_lanes.log
python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp SEGMI_SW_LANES=1 && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inf_lanes -- python3 bench.py --workload infer --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/inf_lanes.log 2>&1
