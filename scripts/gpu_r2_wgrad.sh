#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "wgrad or conv or unet or golden or fused" > gpurun_out/r2/pytest_wgrad.log 2>&1
rc=$?
tail -8 gpurun_out/r2/pytest_wgrad.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/wgrad_bench.py > gpurun_out/r2/wgrad_ws1.log 2>&1 && cat gpurun_out/r2/wgrad_ws1.log || { tail -5 gpurun_out/r2/wgrad_ws1.log; exit 1; }
export SEGMI_LIB=$PWD/segmantic_amd/csrc/libsegmi_diag.so
for w in top toptf; do
for d in 0 1 4 5; do
  SEGMI_WGRAD_DBG=$d timeout -k 10 100 python scripts/wgrad_one.py $w 5 2>&1 | grep -v amdgpu.ids
done; done
unset SEGMI_LIB
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_ws1.json 2> gpurun_out/r2/bench.err && python -c "
import json;d=json.load(open('gpurun_out/r2/bench_ws1.json'));print('ms/step', d['ms_per_step'], 'top fwd ms', d['roofline']['avg_launch_ms'])" || exit 1
