#!/bin/bash
# per-kernel register / LDS / occupancy summary of one .hip file (compiler view):  tools/kres.sh hfnet_slam_amd/csrc/kernels_block.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -fPIC -x hip -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -v "is not a recognized feature" | python3 -c '
import sys,re,subprocess
cur={}
rows=[]
for l in sys.stdin:
    m=re.search(r"remark: (?:\s*)([A-Za-z ]+): (.+?) \[-Rpass", l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name":
        if cur: rows.append(cur)
        cur={"name":v}
    else: cur[k]=v
if cur: rows.append(cur)
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=re.sub(r"\(.*","",n).replace("void hfnet::","").replace("hfnet::","")
    print("%-46s VGPR %4s AGPR %3s spill %3s SGPR %3s occ %2s LDS %6s"%(n[:46],r.get("VGPRs"),r.get("AGPRs"),r.get("VGPRs Spill", r.get("ScratchSize [bytes/lane]")),r.get("TotalSGPRs"),r.get("Occupancy [waves/SIMD]"),r.get("LDS Size [bytes/block]")))
'
