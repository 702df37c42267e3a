#!/bin/bash
mkdir -p gpurun_out/r4
out=gpurun_out/r4/ring4_diag1.txt; : > $out
timeout -k 10 120 python scripts/ring3_debug.py 2>&1 | grep -v amdgpu | cut -c1-200 | grep -v "^   "
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv or ring or bn_bwd or sums or affine" > gpurun_out/r4/ring3_ops.log 2>&1 || { tail -30 gpurun_out/r4/ring3_ops.log; exit 1; }
tail -2 gpurun_out/r4/ring3_ops.log
for d in 0 12 10 8 6; do
  SEGMI_RING3_DBG=$d timeout -k 10 120 python scripts/ring3_diag.py 8 plain fwd_tf dgrad dgrad_sums >> $out 2>&1 || exit 1
done
SEGMI_RING4=0 timeout -k 10 120 python scripts/ring3_diag.py 8 plain fwd_tf dgrad dgrad_sums >> $out 2>&1 || exit 1
grep -v amdgpu.ids $out
