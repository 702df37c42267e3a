#!/bin/bash
export SEGMI_LIB=$PWD/segmantic_amd/csrc/libsegmi_diag.so
for d in 5 69 4 0; do
SEGMI_WGRAD_DBG=$d timeout -k 10 100 python scripts/wgrad_stamps.py top 2>&1 | grep -v amdgpu.ids
done
