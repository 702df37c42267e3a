#!/bin/bash
# inference: per-kernel times with one lane (SEGMI_SERIAL=1), and un-profiled one-lane vs two-lane rate
mkdir -p gpurun_out/r2
for v in 0 1; do
if [ $v = 1 ]; then export SEGMI_SERIAL=1; fi
timeout -k 10 200 python3 bench.py --workload infer --no-cpu-baseline --steps 4 --warmup 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial $v infer', d['value'], d['unit'])" || exit 1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r2/infser
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/infser -- python3 bench.py --workload infer --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2/infser.log 2>&1 || exit 1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2/infser/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms per volume', tot/3e6)
for r in rows[:22]:
    print(f"{r['Name'][:95]:95s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/3e6:8.3f} ms/vol {float(r['AverageNs'])/1e3:9.1f} us")
PY
