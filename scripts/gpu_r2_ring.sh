#!/bin/bash
export SEGMI_LIB=$PWD/segmantic_amd/csrc/libsegmi_diag.so
for d in 0 8 0 8; do SEGMI_RING2_DBG=$d timeout -k 10 100 python scripts/ring2_diag.py 8 2>&1 | grep -v amdgpu; done
unset SEGMI_LIB
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "conv or unet or golden or fused or sliding" 2>&1 | tail -3
