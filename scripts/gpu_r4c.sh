#!/bin/bash
# one-launch slab reduce + bias gradient in the statistics launch: tests, then the step A/B
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "wgrad or bias or stats or dice or train or step or parity or determin" > gpurun_out/r4/c_tests.log 2>&1 || { tail -30 gpurun_out/r4/c_tests.log; exit 1; }
tail -2 gpurun_out/r4/c_tests.log
bash scripts/gpu_ab_step.sh SEGMI_SLAB_REDUCE2 1 0 train 3 || exit 1
