#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "fused_full_resolution" 2>&1 | tail -2
timeout -k 10 120 python scripts/dectop_bench.py 16 2>/dev/null
for d in 1 2; do SEGMI_DECTOP_DBG=$d timeout -k 10 120 python scripts/dectop_bench.py 16 2>/dev/null; done
run() { local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],2), 'vol/s', [(f['lanes'], round(f['value'],2)) for f in d['lanes']['figures']])"
}
for i in 1 2; do
  run "two launches " SEGMI_FUSE_EVAL_TOP=0
  run "fused dectop " SEGMI_FUSE_EVAL_TOP=1
done 2>&1 | tee gpurun_out/r3/dectop_ab.txt
