#!/bin/bash
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --size 160 --classes 32 2>&1 | tail -1 | cut -c1-300
cd /tmp && export TMPDIR=/tmp SEGMI_SERIAL=1 && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --size 160 --classes 32 > gpurun_out/c4.log 2>&1
