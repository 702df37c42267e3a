#!/bin/bash
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "fused_full_resolution or fused_eval_decoder" 2>&1 | grep -E "^E|passed|failed|Error" | head
python scripts/dectop_bench.py 16 2>&1 | grep -v amdgpu
export SEGMI_LIB=$PWD/segmantic_amd/csrc/libsegmi_diag.so
for d in 0 1 2 3 4 15; do SEGMI_DECTOP_DBG=$d timeout -k 10 100 python scripts/dectop_bench.py 16 2>&1 | grep -v amdgpu; done
