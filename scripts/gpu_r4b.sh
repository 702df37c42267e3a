#!/bin/bash
# round 4, second half: small-launch ring A/Bs, inference group size, dice forward time
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "dice or sliding or window or pair or blend or loss" > gpurun_out/r4/b_tests.log 2>&1 || { tail -30 gpurun_out/r4/b_tests.log; exit 1; }
tail -2 gpurun_out/r4/b_tests.log
out=gpurun_out/r4/ring3_small.txt; : > $out
for cfg in "" "SEGMI_RING3_XCD=0" "SEGMI_RING_ZS=2" "SEGMI_RING_ZS=2 SEGMI_RING3_XCD=1" "SEGMI_RING_ZS=4 SEGMI_RING3_XCD=0" "SEGMI_RING3_DBG=12" "SEGMI_RING3_DBG=10" "SEGMI_RING3_DBG=6" "SEGMI_RING3_DBG=8"; do
  env $cfg DIAG_SIZE=64 timeout -k 10 120 python scripts/ring3_diag.py 16 fwd_act plain dgrad_sums 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
cat $out
bash scripts/gpu_ab_step.sh SEGMI_SW_GROUP 4 8 infer 2 || exit 1
bash scripts/gpu_ab_step.sh SEGMI_RING_ZS 1 2 infer 1 || exit 1
bash scripts/gpu_kstats.sh train 10 | grep -i "dice\|total\|ring3" 
