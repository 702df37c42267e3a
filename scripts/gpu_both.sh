#!/bin/bash
# GPU test-suite + both bench lines
python -m pytest tests -x -q -m gpu > gpurun_out/t_both.log 2>&1; tail -2 gpurun_out/t_both.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
