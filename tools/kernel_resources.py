#!/usr/bin/env python
"""Register / LDS / spill report of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python tools/kernel_resources.py decompdiff_amd/csrc/dd_attention2.hip [extra hipcc flags]
"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
KEYS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
        "SGPRs Spill": "sspill", "VGPRs Spill": "vspill", "LDS Size [bytes/block]": "lds"}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|[A-Za-z]+(?: Spill| Size \[bytes/block\]| \[bytes/lane\]| \[waves/SIMD\])?): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:78]}
        rows.append(cur)
    elif cur is not None and k in KEYS:
        cur[KEYS[k]] = v
print(f"{'kernel':78s} vgpr agpr sgpr scratch vspill sspill occ    lds")
for r in rows:
    print(f"{r['name']:78s} {r.get('vgpr','?'):>4} {r.get('agpr','?'):>4} {r.get('sgpr','?'):>4} {r.get('scratch','?'):>7} "
          f"{r.get('vspill','?'):>6} {r.get('sspill','?'):>6} {r.get('occ','?'):>3} {r.get('lds','?'):>6}")
