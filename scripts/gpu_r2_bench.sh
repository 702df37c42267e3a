#!/bin/bash
mkdir -p gpurun_out/r2
timeout -k 10 900 python bench.py "$@" > gpurun_out/r2/bench_all.json 2> gpurun_out/r2/bench_all.err; echo rc=$?; tail -3 gpurun_out/r2/bench_all.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2/bench_all.json"))
print("train ms/step", d["ms_per_step"], "value", d["value"])
r = d.get("roofline") or {}
print("roofline:", r.get("kernel"), "| frac", r.get("frac"), "avg ms", r.get("avg_launch_ms"), "traffic", r.get("traffic"))
print("cpu:", d.get("cpu_baseline"))
i = d.get("infer", {})
print("infer:", {k: v for k, v in i.items() if k not in ("roofline", "config")})
print("infer roofline:", (i.get("roofline") or {}).get("kernel"), (i.get("roofline") or {}).get("frac"), (i.get("roofline") or {}).get("avg_launch_ms"))
print("infer config:", i.get("config"))
f = d.get("f32_parity_mode", {})
print("f32:", {k: v for k, v in f.items() if k not in ("roofline", "what")}, (f.get("roofline") or {}).get("kernel"), (f.get("roofline") or {}).get("frac"))
print("fit:", d.get("fit"))
print("config:", d["config"])
PY
