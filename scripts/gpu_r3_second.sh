#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t2.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t2.log
tail -5 gpurun_out/r3/t2.log
for v in none cpu cpu_threads1 cpu_sleep3; do
  timeout -k 10 200 python scripts/probe_cpu_then_infer.py $v 2>/dev/null | tail -1 | tee -a gpurun_out/r3/probe_cpu.jsonl
done
OMP_WAIT_POLICY=passive timeout -k 10 200 python scripts/probe_cpu_then_infer.py cpu 2>/dev/null | tail -1 | tee -a gpurun_out/r3/probe_cpu.jsonl
GPU_MAX_HW_QUEUES=4 timeout -k 10 200 python scripts/probe_cpu_then_infer.py cpu 2>/dev/null | tail -1 | tee -a gpurun_out/r3/probe_cpu.jsonl
timeout -k 10 400 python bench.py > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3/bench_b.json"))
inf=d.get("infer",{})
print("train ms", round(d["ms_per_step"],3), "infer", inf.get("value"), [(f["lanes"], round(f["value"],2), round(f["host_enqueue_ms_per_volume"],1)) for f in inf.get("lanes",{}).get("figures",[])], "fit", d.get("fit",{}).get("ms_per_step"), d.get("cpu_baseline",{}).get("value"), inf.get("cpu_baseline",{}).get("value"))
PY
