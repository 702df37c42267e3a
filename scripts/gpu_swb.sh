#!/bin/bash
# does a smaller window batch keep the top ConvT output in the 256 MB Infinity Cache for the top conv?
cd /tmp && export TMPDIR=/tmp SEGMI_SW_LANES=1 && cd $GRAFT_REPO_ROOT
for b in 2 1; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inf_swb$b -- python3 bench.py --workload infer --steps 2 --warmup 1 --no-cpu-baseline --sw-batch $b > gpurun_out/inf_swb$b.log 2>&1
tail -1 gpurun_out/inf_swb$b.log | cut -c1-200
done
