#!/bin/bash
# kernel-time summaries (rocprofv3 --kernel-trace --stats) of the training step and of inference: gpurun_out/r4/kstats_*
mkdir -p gpurun_out/r4 gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SEGMI_SW_LANES=${SEGMI_SW_LANES:-1}    # per-kernel times: one inference lane
for wl in ${1:-train infer}; do
  rm -rf gpurun_out/prof/ks_$wl
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ks_$wl -- python3 bench.py --workload $wl --steps ${2:-10} --warmup 3 --no-cpu-baseline --no-lane-ab > gpurun_out/prof/ks_$wl.log 2>&1 || { tail -5 gpurun_out/prof/ks_$wl.log; exit 1; }
  f=$(ls gpurun_out/prof/ks_$wl/*/*kernel_stats.csv | head -1)
  python3 scripts/kstats.py $f > gpurun_out/r4/kstats_$wl.txt 2>&1 || cp $f gpurun_out/r4/kstats_$wl.csv
  head -40 gpurun_out/r4/kstats_$wl.txt
done
