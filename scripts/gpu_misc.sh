#!/bin/bash
echo "f32 128^3 b8 k16"; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision f32 2>&1 | tail -1 | cut -c1-200
echo "bf16 96^3 b8 k16"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --size 96 2>&1 | tail -1 | cut -c1-200
echo "bf16 96^3 b8 k3"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --size 96 --classes 3 2>&1 | tail -1 | cut -c1-200
echo "bf16 128^3 b2 k16"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 2 2>&1 | tail -1 | cut -c1-200
echo "infer 256^3"; python bench.py --workload infer --volume 256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
