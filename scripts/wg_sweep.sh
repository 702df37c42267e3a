#!/bin/bash
cd /tmp && export TMPDIR=/tmp SEGMI_SERIAL=1 SEGMI_WGRAD_NO41=1 && cd $GRAFT_REPO_ROOT
for t in 256 768; do
  export SEGMI_WGRAD_WGS3=$t
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/wg3_$t -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/wg3_$t.log 2>&1 || exit 1
done
