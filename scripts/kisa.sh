#!/bin/bash
# usage: kisa.sh <file.hip> <kernel-name-substring>   -- resource usage + MFMA-region instruction mix
cd /root/repo/segmantic_amd/csrc || exit 1
f=$1; k=$2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-gpu-rdc -Rpass-analysis=kernel-resource-usage -S --cuda-device-only $f -o /tmp/kisa.s 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|Occupancy" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/^.*remark: *//' | paste - - - - - | grep -E "error|$k" | head -12
python3 - "$k" <<'PY'
import re,collections,sys
s=open('/tmp/kisa.s').read()
k=sys.argv[1]
names=[m.group(1) for m in re.finditer(r"\n(_Z\w+):", s) if k in m.group(1)]
for name in names[:2]:
    i0=s.index("\n"+name+":"); i1=s.index("s_endpgm",i0)
    lines=[l.split(";")[0].strip() for l in s[i0:i1].split("\n")]
    lines=[l for l in lines if l and (not l.startswith(".") or l.startswith(".L"))]
    idx=[i for i,l in enumerate(lines) if l.startswith("v_mfma")]
    if not idx: continue
    lo,hi=idx[0],idx[-1]
    c=collections.Counter(l.split()[0] for l in lines[lo-30:hi+1])
    print(name[:90]); print(len(lines),"lines;",len(idx),"mfma; region",hi-lo, c.most_common(14))
PY
