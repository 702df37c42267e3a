#!/bin/bash
# kernel-time summary of the BASELINE config 4 step on one GPU (160^3, 32 labels, batch 4)
mkdir -p gpurun_out/r4 gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/ks_c4
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ks_c4 -- python3 bench.py --workload train --size 160 --classes 32 --batch ${1:-4} --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof/ks_c4.log 2>&1 || { tail -5 gpurun_out/prof/ks_c4.log; exit 1; }
python3 scripts/kstats.py $(ls gpurun_out/prof/ks_c4/*/*kernel_stats.csv | head -1) 7 30 > gpurun_out/r4/kstats_c4.txt
cut -c1-150 gpurun_out/r4/kstats_c4.txt
