#!/bin/bash
mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1234; do
rm -rf gpurun_out/r2/rp_$v
SEGMI_RING2_DBG=$v timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r2/rp_$v -- python3 scripts/ring2_diag.py 8 > gpurun_out/r2/rp_$v.log 2>&1
python3 - $v <<'PY'
import csv,glob,sys
v=sys.argv[1]
f=glob.glob(f'gpurun_out/r2/rp_{v}/**/*counter_collection.csv',recursive=True)[0]
vals=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'conv_ring2' in r['Kernel_Name']]
print('dbg',v,'FETCH_SIZE KB avg',sum(vals)/len(vals),'n',len(vals))
PY
grep "dbg=" gpurun_out/r2/rp_$v.log
done
