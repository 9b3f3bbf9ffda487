#!/usr/bin/env python3
"""Timeline of the decode kernels from a rocprofv3 --kernel-trace CSV: per consecutive pair (by start time) how much the
younger kernel's residency overlaps the older one's, plus per-instantiation duration stats.
Usage: python tools/trace_overlap.py <dir-with-*kernel_trace.csv> [n_last_dispatches]"""
import collections
import csv
import glob
import re
import sys

src = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 512
f = glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(r"qllm::strip\d?_kernel", r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
prev = None
ov_tot = span = 0
print("  start_us    dur_us  overlap_with_prev_us  gap_us  queue/stream  grid  kernel")
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.split(r"strip\d?_kernel", r["Kernel_Name"])[1].split("(")[0][:44]
    ov = gap = 0.0
    if prev is not None:
        ov = max(0, min(prev[1], e) - s) / 1e3
        gap = max(0, s - prev[1]) / 1e3
        ov_tot += ov
    if i < 40:
        print(f"{(s - t0) / 1e3:10.2f} {(e - s) / 1e3:9.2f} {ov:12.2f} {gap:14.2f}   q{r.get('Queue_Id', '?')}/s{r.get('Stream_Id', '?')}  "
              f"{r.get('Grid_Size_X', r.get('Grid_Size', '?')):>7} {name}")
    prev = (s, max(e, prev[1]) if prev else e)
span = (max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e3
dur = collections.defaultdict(list)
for r in rows:
    dur[(re.split(r"strip\d?_kernel", r["Kernel_Name"])[1].split("(")[0][:44], r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"\n{len(rows)} dispatches over {span:.1f} us: sum of durations {sum(sum(v) for v in dur.values()):.1f} us, "
      f"pairwise overlap {ov_tot:.1f} us")
for k, v in sorted(dur.items()):
    v.sort()
    print(f"  {k[0]:46s} grid {k[1]:>7}  n={len(v):4d}  min {v[0]:6.2f}  median {v[len(v) // 2]:6.2f}  mean {sum(v) / len(v):6.2f} us")
