#!/bin/bash
# full GPU test-suite, bench line, and a serial (single-stream) kernel-stats profile
tag=${1:-x}
python -m pytest tests -x -q -m gpu > gpurun_out/t_$tag.log 2>&1; tail -3 gpurun_out/t_$tag.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
cd /tmp && export TMPDIR=/tmp SEGMI_SERIAL=1 && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ser_$tag -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/ser_$tag.log 2>&1
