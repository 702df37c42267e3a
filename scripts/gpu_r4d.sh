#!/bin/bash
mkdir -p gpurun_out/r4
bash scripts/gpu_ab_step.sh SEGMI_CARRY_LEVELS 1 2 train 2 || exit 1
bash scripts/gpu_ab_step.sh SEGMI_CARRY_LEVELS 3 4 train 2 || exit 1
