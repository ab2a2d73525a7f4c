#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/scratch/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
here = os.path.dirname(os.path.abspath(__file__))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-fno-slp-vectorize",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/sba_ru.so", os.path.join(here, "sba_api.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: ([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        if cur: rows.append(cur)
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void sba::", "")}
    else:
        cur[k] = v
if cur: rows.append(cur)
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    print(f"{r['name'][:44]:44s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} "
          f"{r.get('ScratchSize','?'):>8s} {r.get('LDS Size','?'):>7s} {r.get('Occupancy','?'):>4s}")
