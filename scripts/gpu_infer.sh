#!/bin/bash
# inference bench line + single-stream kernel stats
tag=${1:-x}
python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp SEGMI_SW_LANES=1 && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inf_$tag -- python3 bench.py --workload infer --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/inf_$tag.log 2>&1
