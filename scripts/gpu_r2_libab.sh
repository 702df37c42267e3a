#!/bin/bash
# A/B of two builds inside one box: libsegmi.so (new) vs libsegmi_old.so (previous commit), alternating
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
L=$GRAFT_REPO_ROOT/segmantic_amd/csrc
for v in new old new old new old; do
lib=$L/libsegmi.so; [ $v = old ] && lib=$L/libsegmi_old.so
SEGMI_LIB=$lib timeout -k 10 200 python3 bench.py --workload train --no-cpu-baseline --steps 30 --warmup 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v ms_per_step', d['ms_per_step'], 'top conv ms', d['roofline']['avg_launch_ms'])" || exit 1
done
