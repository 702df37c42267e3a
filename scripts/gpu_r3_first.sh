#!/bin/bash
# round 3, first GPU call: the whole GPU suite, the default bench line, and the hardware-queue A/B
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/t1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/t1.log
tail -5 gpurun_out/r3/t1.log
timeout -k 10 400 python bench.py > gpurun_out/r3/bench_q8.json 2> gpurun_out/r3/bench_q8.err; echo "bench q8 rc=$?"
GPU_MAX_HW_QUEUES=4 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_q4.json 2> gpurun_out/r3/bench_q4.err; echo "bench q4 rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_q8b.json 2> gpurun_out/r3/bench_q8b.err; echo "bench q8b rc=$?"
GPU_MAX_HW_QUEUES=4 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_q4b.json 2> gpurun_out/r3/bench_q4b.err; echo "bench q4b rc=$?"
python - <<'PY'
import json
for n in ("q8","q4","q8b","q4b"):
    try:
        d=json.load(open(f"gpurun_out/r3/bench_{n}.json"))
        inf=d.get("infer",{})
        print(n, "train ms", round(d["ms_per_step"],3), "infer", inf.get("value"), [(f["lanes"], round(f["value"],2), [round(x,1) for x in f["lane_busy_ms_per_volume"]], round(f["host_enqueue_ms_per_volume"],1)) for f in inf.get("lanes",{}).get("figures",[])], "fit", d.get("fit",{}).get("ms_per_step"))
    except Exception as e:
        print(n, "ERR", e)
PY
