#!/usr/bin/env python3
"""Per-kernel register use of one .hip file, from hipcc's -Rpass-analysis=kernel-resource-usage remarks (no GPU needed).
Usage: python tools/kres.py qllm_amd/csrc/strip_sm.hip [filter-substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"^void qllm::", "", cur)
        cur = re.sub(r"\(.*$", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':90s} VGPR AGPR spill scratch occ  SGPR  LDS")
for k, v in rows.items():
    if flt and flt not in k:
        continue
    print(f"{k:90s} {v.get('VGPRs', -1):4d} {v.get('AGPRs', -1):4d} {v.get('VGPRs Spill', -1):5d} {v.get('ScratchSize [bytes/lane]', -1):7d} "
          f"{v.get('Occupancy [waves/SIMD]', -1):3d} {v.get('TotalSGPRs', -1):5d} {v.get('LDS Size [bytes/block]', -1):5d}")
