#!/bin/bash
# A/B of the training step / inference between two settings of one environment variable, alternating runs in one call
# usage: gpu_ab_step.sh VAR A B [workload] [reps]
var=$1; a=$2; b=$3; wl=${4:-train}; reps=${5:-2}
mkdir -p gpurun_out/r4
for i in $(seq 1 $reps); do
  for v in $a $b; do
    env $var=$v timeout -k 10 300 python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-lane-ab > gpurun_out/r4/ab_${var}_${v}_$i.json 2>/dev/null || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/r4/ab_${var}_${v}_$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$var=$v run $i $wl", round(d.get("ms_per_step",0),3), "ms/step; value", round(d["value"],3), "; top", round(r["avg_launch_ms"],4), "ms", r["kernel"][:48])
PY
  done
done
