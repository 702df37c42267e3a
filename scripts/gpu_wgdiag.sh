#!/bin/bash
for t in 256 512 768 1024; do echo "target $t"; SEGMI_WGRAD_T11=$t python scripts/wgrad_diag.py 8 2>&1 | tail -1; done
