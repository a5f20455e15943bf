#!/bin/bash
# kernel resource table of one HIP source: tools/kres.sh openvo_amd/csrc/sgbm.hip [name filter]
src=$1; filt=${2:-.}
cd "$(dirname "$src")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Rpass-analysis=kernel-resource-usage -c "$(basename "$src")" -o /tmp/kres.o 2>&1 |
python3 -c '
import sys,re,subprocess
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r"remark: (?:[^:]*:\d+:\d+: )?\s*(.*?) \[-Rpass",l) or re.search(r":\d+:\d+:\s+(.*?) \[-Rpass",l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur={"name":t.split(":",1)[1].strip()};rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.split(":",1);cur[k.strip()]=v.strip()
names=[r["name"] for r in rows]
dem=subprocess.run(["c++filt"],input="\n".join(names),capture_output=True,text=True).stdout.split("\n")
for r,d in zip(rows,dem):
    d=re.sub(r"\(.*","",d).replace("void ","")
    print("%-60s vgpr %4s agpr %3s sgpr %4s scratch %4s occ %2s spillV %3s spillS %3s"%(d[:60],r.get("VGPRs"),r.get("AGPRs"),r.get("SGPRs"),r.get("ScratchSize [bytes/lane]"),r.get("Occupancy [waves/SIMD]"),r.get("VGPRs Spill"),r.get("SGPRs Spill")))
' | grep -E "$filt"
