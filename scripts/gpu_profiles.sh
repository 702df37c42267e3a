#!/bin/bash
# profile set (round 2 layout, reused in rounds 3 and 4): the driver-style bench line, kernel stats (overlapped run), per-kernel serial
# trace, PMC FETCH/WRITE passes (serial) for training; kernel stats + PMC for inference
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
python bench.py --steps 20 --warmup 5 > gpurun_out/prof/bench_all.json 2> gpurun_out/prof/bench_all.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SEGMI_SW_LANES=1    # per-kernel profiles: one lane (the library default of 3 runs copies of a launch side by side)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/train_stats -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof/train_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/infer_stats -- python3 bench.py --workload infer --steps 2 --warmup 1 --no-cpu-baseline --no-lane-ab > gpurun_out/prof/infer_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/inf_fetch -- python3 bench.py --workload infer --steps 1 --warmup 1 --no-cpu-baseline --no-lane-ab > gpurun_out/prof/inf_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/inf_write -- python3 bench.py --workload infer --steps 1 --warmup 1 --no-cpu-baseline --no-lane-ab > gpurun_out/prof/inf_write.log 2>&1 || exit 1
export SEGMI_SERIAL=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/ser_time -- python3 bench.py --workload train --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof/ser_time.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/ser_fetch -- python3 bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof/ser_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/ser_write -- python3 bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof/ser_write.log 2>&1 || exit 1
echo done
