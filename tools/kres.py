#!/usr/bin/env python3
"""Print per-kernel register/LDS/scratch usage of a HIP source (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude",
       "-Ietol_amd/csrc", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+([\w \[\]/]+?): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ("Function Name", "Name"):
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    print(f"{name[:90]:90s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('TotalSGPRs', r.get('SGPRs','?')):>4} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?'):>2} lds={r.get('LDS Size [bytes/block]','?')}")
