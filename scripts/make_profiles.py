#!/usr/bin/env python3
"""Turn the raw output of scripts/gpu_profiles.sh (gpurun_out/prof/) into the committed summaries under
profiles/ (tag = round, default r01).  usage: make_profiles.py [tag]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
OUT = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(os.path.join(SRC, pattern))
    if not files:
        raise SystemExit(f"missing {pattern}")
    return max(files, key=os.path.getmtime)


def run(script, *args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), *args], check=True,
                          capture_output=True, text=True).stdout


for wl, name in (("train", "train_b8_128_bf16"), ("infer", "infer_512_bf16")):
    st = newest(f"{wl}_stats/*/*kernel_stats.csv")
    shutil.copy(st, os.path.join(OUT, f"{tag}_{name}_kernel_stats.csv"))
    steps = "13"
    if wl == "infer":
        # volumes of the run (the warm-up is adaptive since round 3) = launches of the fused decoder top / 22 window
        # groups per 512^3 volume (343 windows in groups of 16); the blend runs as 7 z-slab launches per volume
        calls = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(st))}
        dt = sum(v for k, v in calls.items() if "dectop_kernel" in k)
        steps = str(dt // 22 if dt and dt % 22 == 0 else sum(v for k, v in calls.items() if "sw_blend" in k) // 7)
    open(os.path.join(OUT, f"{tag}_{name}_kernel_stats.txt"), "w").write(
        f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {'--workload infer --steps 2 --warmup 1' if wl == 'infer' else '--workload train --steps 10 --warmup 3'} --no-cpu-baseline\n"
        f"# ({'both streams overlapped' if wl == 'train' else 'one lane (SEGMI_SW_LANES=1 pinned for per-kernel times; the library default is 3)'}; every launch of the run = {steps} steps incl. warm-up; setup kernels included)\n"
        + run("kstats.py", st, steps))
shutil.copy(os.path.join(SRC, "bench_all.json"), os.path.join(OUT, f"{tag}_bench_all.json"))

t = newest("ser_time/*/*kernel_trace.csv")
f = newest("ser_fetch/*/*counter_collection.csv")
w = newest("ser_write/*/*counter_collection.csv")
open(os.path.join(OUT, f"{tag}_train_bw_table.txt"), "w").write(run("bw_table.py", t, f, w))
open(os.path.join(OUT, f"{tag}_train_b8_128_bf16_serial_last_step_by_grid.txt"), "w").write(run("grid_table.py", t))
open(os.path.join(OUT, f"{tag}_train_b8_128_bf16_last_step_by_grid.txt"), "w").write(
    run("grid_table.py", newest("train_stats/*/*kernel_trace.csv")))


def launches(path, substr, grid):
    """counter values of the dispatches of a kernel (name substring, total threads) in time order"""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    return [(float(r["Counter_Value"]), r["Kernel_Name"]) for r in rows
            if substr in r["Kernel_Name"] and int(r["Grid_Size"]) == grid]


sys.path.insert(0, ROOT)
import bench  # noqa: E402  (csrc_hash: stamps the measurement to the kernel sources it was taken on)

# training: forward launch of the full-resolution 16 -> 16 conv = the MODE 0 (plain) ring3 dispatch of 1024
# workgroups of a step (the input-gradient launch of that layer is MODE 4); batch 8 x 128^3
fk, kname = launches(f, "conv_ring3_kernel<0>", 262144)[-1]
wk, _ = launches(w, "conv_ring3_kernel<0>", 262144)[-1]
alg = 8 * 128 ** 3 * 16 * 2 * 2
# inference: the same layer on a full group of 16 windows (2048 workgroups)
fi = newest("inf_fetch/*/*counter_collection.csv")
wi = newest("inf_write/*/*counter_collection.csv")
# (round 3: the fused decoder top, csrc/dectop.hip: 1024 workgroups of 512 threads per group of 16 windows)
fki = [v for v, _ in launches(fi, "dectop_kernel", 524288)]
wki = [v for v, _ in launches(wi, "dectop_kernel", 524288)]
fki, wki = sorted(fki)[len(fki) // 2], sorted(wki)[len(wki) // 2]
alg_i = 16 * (64 ** 3 * 32 + 128 ** 3 * 16) * 2
json.dump({
    "source_hash": bench.csrc_hash(),
    "kernel": kname + " (training: full-resolution 16->16 k3 conv forward with identity residual from LDS; "
              "inference: dectop_kernel, the fused ConvTranspose3d 32->16 + conv 16->16 of the decoder top)",
    "command": "SEGMI_SERIAL=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py "
               "--workload train --steps 1 --warmup 1 --no-cpu-baseline ; same with --pmc WRITE_SIZE (separate passes); "
               "inference: the same two passes of --workload infer (scripts/gpu_profiles.sh)",
    "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced streaming reads (MI355X_MICROARCH.md HBM section) "
                  "-> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
    "train_FETCH_SIZE_KB": fk, "train_WRITE_SIZE_KB": wk,
    "train_top_conv_fwd_hbm_bytes_per_launch": (2 * fk + wk) * 1024, "train_algorithmic_bytes_per_launch": alg,
    "infer_FETCH_SIZE_KB": fki, "infer_WRITE_SIZE_KB": wki,
    "infer_top_conv_fwd_hbm_bytes_per_launch": (2 * fki + wki) * 1024, "infer_algorithmic_bytes_per_launch": alg_i,
    "note": "reads exceed the input by the (8+2)x(16+2)/(8x16) y/x halo of neighbouring columns that miss the XCD's L2 "
            "(round 4: the 1024-workgroup launch no longer uses the XCD-aware column map -- it was 10 % slower with it -- "
            "so most of the halo is re-fetched past L2; FETCH_SIZE counts Infinity-Cache hits too, MI355X_MICROARCH.md); "
            "writes are exact",
}, open(os.path.join(OUT, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("profiles written:", sorted(os.listdir(OUT)))
