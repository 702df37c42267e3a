#!/bin/bash
mkdir -p gpurun_out/r3
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload train --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],3))"
}
for i in 1 2 3; do
  run "fused BN 256 WGs, defer       " SEGMI_BN_FUSED_WGS=256
  run "fused BN 128 WGs, defer       " X=1
  run "fused BN 128 WGs, no defer    " SEGMI_DEFER_TOP_WGRAD=0
  run "fused BN  64 WGs, defer       " SEGMI_BN_FUSED_WGS=64
  run "BN small as two launches      " SEGMI_FUSE_BN_BWD_SMALL=0
done 2>&1 | tee gpurun_out/r3/sched2_ab.txt
timeout -k 10 120 python scripts/fused_bn_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3/fused_bn_bench.txt
