#!/usr/bin/env python3
"""Register / LDS / scratch figures of every kernel in a built libsegmi.so (code-object metadata).
usage: kernel_meta.py [lib.so] [name-filter]"""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

llvm = "/opt/rocm/lib/llvm/bin"
so = Path(sys.argv[1] if len(sys.argv) > 1 else "segmantic_amd/csrc/libsegmi.so").resolve()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    shutil.copy(so, td / "lib.so")
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", "lib.so"], cwd=td, check=True, capture_output=True)
    rows = []
    for f in sorted(td.glob("lib.so.*gfx950")):
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", str(f)], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
            name = g("name").group(1)
            rows.append((name, int(g("vgpr_count").group(1)), int(blk.split()[0]), int(g("sgpr_count").group(1)),
                         int(g("group_segment_fixed_size").group(1)), int(g("private_segment_fixed_size").group(1)),
                         int(g("vgpr_spill_count").group(1)), int(g("kernarg_segment_size").group(1))))
    for r in sorted(rows):
        if flt in r[0]:
            dem = subprocess.run(["c++filt", r[0]], capture_output=True, text=True).stdout.strip()
            print(f"vgpr {r[1]:3d} agpr {r[2]:3d} sgpr {r[3]:3d} lds {r[4]:6d} scratch {r[5]:4d} spill {r[6]:3d} kernarg {r[7]:4d}  {dem[:110]}")
