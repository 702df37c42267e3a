#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t7.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t7.log
tail -4 gpurun_out/r3/t7.log
grep -q "pytest rc=0" gpurun_out/r3/t7.log || exit 1
timeout -k 10 120 python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 500 python bench.py > gpurun_out/r3/bench_c.json 2> gpurun_out/r3/bench_c.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3/bench_c.json"))
inf=d.get("infer",{})
print("train ms", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), "infer", round(inf.get("value",0),2), [(f["lanes"], round(f["value"],2), round(f["host_enqueue_ms_per_volume"],1)) for f in inf.get("lanes",{}).get("figures",[])], "infer roofline", inf.get("roofline",{}).get("frac"), "fit", d.get("fit",{}).get("ms_per_step"), "f32", d.get("f32_parity_mode",{}).get("ms_per_step"))
print(inf.get("roofline",{}).get("kernel","")[:200])
PY
