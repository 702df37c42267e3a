#!/bin/bash
# where the waves of the step's big kernels spend their cycles (SQ counters, serial run) + the effective clock
set -o pipefail
mkdir -p gpurun_out/r4 gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/sq1 gpurun_out/prof/sq2
export SEGMI_SERIAL=1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof/sq1 -- python3 bench.py --workload train --steps 1 --warmup 2 --no-cpu-baseline > gpurun_out/prof/sq1.log 2>&1 || exit 1
python scripts/pmc_top.py $(ls -t gpurun_out/prof/sq1/*/*counter_collection.csv | head -1) 28 > gpurun_out/r4/sq_top.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/prof/sq2 -- python3 bench.py --workload train --steps 1 --warmup 2 --no-cpu-baseline > gpurun_out/prof/sq2.log 2>&1 || exit 1
python - <<'PY' > gpurun_out/r4/sq_top2.txt
import csv, glob, collections, os
f = sorted(glob.glob("gpurun_out/prof/sq2/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    e = d.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"][:58], "dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
    e["c"][r["Counter_Name"]] = float(r["Counter_Value"])
for e in sorted(d.values(), key=lambda e: -e["dur"])[:28]:
    c = e["c"]
    print(f"{e['name']:58s} {e['dur']:8.1f}us " + " ".join(f"{k[3:] if k.startswith('SQ_') else k}={v:.3g}" for k, v in c.items()))
PY
tail -3 gpurun_out/r4/sq_top.txt
