#!/bin/bash
# time split of the top ring conv (diag build): 1 = no staging loads, 2 = no stores, 4 = no MFMA loop,
# 8 = no commit, 16 = no epilogue, 32 = no end-of-step barrier
export SEGMI_LIB=$GRAFT_REPO_ROOT/segmantic_amd/csrc/libsegmi_diag.so
for v in 0 7 15 23 31 63; do
SEGMI_RING2_DBG=$v timeout -k 10 120 python3 scripts/ring2_diag.py 8 || exit 1
done
