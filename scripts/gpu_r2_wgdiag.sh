#!/bin/bash
mkdir -p gpurun_out/r2
export SEGMI_LIB=$PWD/segmantic_amd/csrc/libsegmi_diag.so
for w in top toptf; do
for d in 0 1 4 5; do
  SEGMI_WGRAD_DBG=$d timeout -k 10 100 python scripts/wgrad_one.py $w 5 2>&1 | grep -v amdgpu.ids
done; done
unset SEGMI_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-20)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r2/pmc_$n -- python3 scripts/wgrad_one.py top 2 > gpurun_out/r2/pmc_$n.log 2>&1 || { tail -5 gpurun_out/r2/pmc_$n.log; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r2/pmc_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'wgrad' in r['Kernel_Name']:
                acc[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(k, 'n=%d' % len(v), 'avg=%.4g' % (sum(v) / len(v)))
PY
