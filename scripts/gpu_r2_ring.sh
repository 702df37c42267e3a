#!/bin/bash
# time split of the top ring conv (diag build): 1 = no staging loads, 2 = no stores, 4 = no MFMA loop
export SEGMI_LIB=$GRAFT_REPO_ROOT/segmantic_amd/csrc/libsegmi_diag.so
for v in 0 1 2 4 3 5 6 7; do
SEGMI_RING2_DBG=$v timeout -k 10 120 python3 scripts/ring2_diag.py 8 || exit 1
done
