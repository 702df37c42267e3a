#!/bin/bash
python -m pytest tests -x -q -m gpu > gpurun_out/t_pair.log 2>&1; tail -3 gpurun_out/t_pair.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp SEGMI_SW_LANES=1 && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inf_pair -- python3 bench.py --workload infer --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/inf_pair.log 2>&1
