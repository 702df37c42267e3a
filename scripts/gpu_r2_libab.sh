#!/bin/bash
# A/B of two builds inside one box: libsegmi.so (new) vs libsegmi_old.so (previous commit), alternating
mkdir -p gpurun_out/r2
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r2/t_all.log 2>&1; tail -2 gpurun_out/r2/t_all.log
grep -q " passed" gpurun_out/r2/t_all.log || exit 1
L=$GRAFT_REPO_ROOT/segmantic_amd/csrc
for v in new old new old new old; do
lib=$L/libsegmi.so; [ $v = old ] && lib=$L/libsegmi_old.so
SEGMI_LIB=$lib timeout -k 10 200 python3 bench.py --workload train --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r2/ab.log 2>&1 || { tail -5 gpurun_out/r2/ab.log; exit 1; }
python3 -c "import sys,json; d=json.loads(open('gpurun_out/r2/ab.log').read().strip().splitlines()[-1]); print('$v ms_per_step', d['ms_per_step'], 'top conv ms', d['roofline']['avg_launch_ms'])"
done
for v in new old new old; do
lib=$L/libsegmi.so; [ $v = old ] && lib=$L/libsegmi_old.so
SEGMI_LIB=$lib timeout -k 10 200 python3 bench.py --workload infer --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r2/ab.log 2>&1 || { tail -5 gpurun_out/r2/ab.log; exit 1; }
python3 -c "import sys,json; d=json.loads(open('gpurun_out/r2/ab.log').read().strip().splitlines()[-1]); print('$v infer', d['value'])"
done
for v in new old new old; do
lib=$L/libsegmi.so; [ $v = old ] && lib=$L/libsegmi_old.so
SEGMI_LIB=$lib timeout -k 10 120 python3 scripts/ring2_diag.py 8 | sed "s/^/$v /" || exit 1
done
