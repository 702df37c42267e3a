#!/bin/bash
mkdir -p gpurun_out/r4 gpurun_out/prof
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | head -30
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/ks_f32
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ks_f32 -- python3 bench.py --workload train --precision f32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof/ks_f32.log 2>&1 || { tail -5 gpurun_out/prof/ks_f32.log; exit 1; }
python3 scripts/kstats.py $(ls gpurun_out/prof/ks_f32/*/*kernel_stats.csv | head -1) 7 24 > gpurun_out/r4/kstats_f32.txt
cat gpurun_out/r4/kstats_f32.txt | cut -c1-160
