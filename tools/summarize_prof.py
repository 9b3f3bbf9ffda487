#!/usr/bin/env python3
"""Condense a rocprofv3 run directory (gpurun_out/prof_rNN) into the small files committed under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r01"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
out = "profiles"
os.makedirs(out, exist_ok=True)

rows = list(csv.DictReader(open(glob.glob(f"{src}/trace/*kernel_stats.csv")[0])))
with open(f"{out}/{tag}_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        if "qllm::" in r["Name"]:
            w.writerow(r)

tr = list(csv.DictReader(open(glob.glob(f"{src}/trace/*kernel_trace.csv")[0])))
d = collections.defaultdict(list)
for r in tr:
    if "qllm::strip_kernel" in r["Kernel_Name"] or "qllm::skinny_kernel" in r["Kernel_Name"] or "qllm::gemm_kernel" in r["Kernel_Name"]:
        d[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = {}
for kind in ("fetch", "write"):
    fs = glob.glob(f"{src}/pmc_{kind}/*counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        if "qllm::strip_kernel" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]))
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    pmc[kind] = {k: v[1] / v[0] for k, v in agg.items()}

bench = json.loads(open(f"{src}/bench_under_rocprof.json").read().strip().splitlines()[-1])
lines = [f"# {tag}: rocprofv3 summary of `python bench.py --steps 20 --warmup 3 --no-extra`", "",
         "Command (on the GPU box, `cd /tmp && export TMPDIR=/tmp` first):",
         "`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 3 --no-extra`;",
         "HBM counters from two separate passes `--kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (5 steps).", "",
         f"bench line under the profiler: value={bench['value']} {bench['unit']}, ms_per_step={bench['ms_per_step']}, "
         f"roofline.achieved={bench['roofline']['achieved']} GB/s (avg launch {bench['roofline']['avg_launch_us']} us incl. gaps)", "",
         "| kernel instance | grid threads | block | launches | min us | median us | avg us | FETCH_SIZE KB/launch (raw) | x2 (gfx950 correction) MB | WRITE_SIZE KB/launch |",
         "|---|---|---|---|---|---|---|---|---|---|"]
for k, v in sorted(d.items()):
    v.sort()
    f = pmc.get("fetch", {}).get((k[0], k[1]))
    wv = pmc.get("write", {}).get((k[0], k[1]))
    lines.append(f"| `{k[0]}` | {k[1]} | {k[2]} | {len(v)} | {v[0] / 1e3:.2f} | {v[len(v) // 2] / 1e3:.2f} | {sum(v) / len(v) / 1e3:.2f} | "
                 f"{f:.0f} | {2 * f * 1024 / 1e6:.1f} | {wv:.1f} |" if f is not None else
                 f"| `{k[0]}` | {k[1]} | {k[2]} | {len(v)} | {v[0] / 1e3:.2f} | {v[len(v) // 2] / 1e3:.2f} | {sum(v) / len(v) / 1e3:.2f} | - | - | - |")
ALG = {98304: ("q/k/v (one grouped launch, 192 strips of 64 columns)", 3 * 8732672),
       176128: ("gate/up (one grouped launch, 344 strips of 64 columns)", 2 * 23455232),
       262144: None}
lines += ["", "Template arguments: <waves per block, columns per lane, k-steps per round, k-steps per group, staged x chunks per lane, "
          "bits, register-A, bf16, row tiles>.  Grid -> launch (Llama-2-7B decode, grouped launches): 98,304 threads = q/k/v, "
          "176,128 = gate/up, `<16,1,8,..>` = o_proj (256 strips of 16 columns, 8.73 MB algorithmic), `<16,1,24,..>` = down_proj "
          "(K = 11008, 23.46 MB).", "",
          "| launch | algorithmic MB | median us | achieved TB/s (algorithmic bytes / median) | HBM traffic MB (FETCH x2 + WRITE) | traffic / algorithmic |",
          "|---|---|---|---|---|---|"]
per_launch = {}
tot_alg = tot_traffic = 0.0
for k, v in sorted(d.items()):
    if k[1] in (98304, 176128):
        name, alg = ALG[k[1]]
    elif "<16, 1, 8," in k[0]:
        name, alg = "o_proj", 8732672
    elif "<16, 1, 24," in k[0]:
        name, alg = "down_proj", 23455232
    else:
        continue
    f = pmc.get("fetch", {}).get((k[0], k[1]))
    wv = pmc.get("write", {}).get((k[0], k[1]))
    med = v[len(v) // 2] / 1e3
    traffic = (2 * f + wv) * 1024 if f is not None and wv is not None else None
    per_launch[name.split(" ")[0]] = traffic
    tot_alg += alg
    tot_traffic += traffic or 0.0
    lines.append(f"| {name} | {alg / 1e6:.2f} | {med:.2f} | {alg / med / 1e6:.2f} | " +
                 (f"{traffic / 1e6:.1f} | {traffic / alg:.3f} |" if traffic else "- | - |"))
lines += ["", "FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM section): doubled. "
          "Traffic within 2-10 % of the algorithmic bytes => no wasted re-reads (the 64-byte-segment strips of o_proj / down_proj "
          "pay the larger margin)."]
open(f"{out}/{tag}_bench_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if tot_traffic:
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 5 "
                         f"--warmup 2 --no-extra; see profiles/{tag}_bench_summary.md",
               "correction": "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request for wide coalesced reads, MI355X_MICROARCH.md "
                             "HBM section); unit KB -> bytes x1024; WRITE_SIZE added",
               "bytes_per_launch_avg_over_step": int(tot_traffic / 4), "per_launch": {k: int(v) for k, v in per_launch.items() if v},
               "algorithmic_bytes_per_launch_avg": int(tot_alg / 4)}, open(f"{out}/{tag}_pmc.json", "w"), indent=2)

# ---- prefill kernels + SQ counters --------------------------------------------------------------------------------------
pf = glob.glob(f"{src}/prefill/*kernel_stats.csv")
if pf:
    rows = [r for r in csv.DictReader(open(pf[0])) if "qllm::" in r["Name"]]
    with open(f"{out}/{tag}_prefill_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    pl = ["# " + tag + ": prefill kernels under rocprofv3", "",
          "`rocprofv3 --kernel-trace --stats --output-format csv -- python tools/kbench.py --m 2048 --iters 40 --layouts GPTQ GEMM`",
          "(M = 2048, Llama-2-7B shapes; per-kernel stats in `" + tag + "_prefill_kernel_stats.csv`, whose average mixes the three shapes).",
          "Per-shape throughput printed by the same run (graph replay, HIP events):", "", "```"]
    pl += [l.rstrip() for l in open(f"{src}/prefill_kbench.log") if l.startswith(("GPTQ", "GEMM"))]
    pl += ["```", ""]
    tr2 = glob.glob(f"{src}/prefill/*kernel_trace.csv")
    if tr2:
        dd = collections.defaultdict(list)
        for r in csv.DictReader(open(tr2[0])):
            if "qllm::gemm2_kernel" in r["Kernel_Name"]:
                dd[(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        pl += ["| kernel | grid threads (tiles x 512) | launches | min us | median us |", "|---|---|---|---|---|"]
        for k, v in sorted(dd.items()):
            v.sort()
            pl.append(f"| `{k[0]}` | {k[1]} | {len(v)} | {v[0] / 1e3:.1f} | {v[len(v) // 2] / 1e3:.1f} |")
        pl.append("")
    ref = f"gpurun_out/{tag}_hipblaslt_ref.log"
    if os.path.exists(ref):
        pl += ["Vendor dense GEMM on the same shape, same run (`tools/one_shape.py --ref`, eager launches, HIP events):", "", "```"]
        pl += [l.rstrip() for l in open(ref) if "TFLOP" in l] + ["```", ""]
    for i in (1, 2):
        sq = f"gpurun_out/pmc_{tag}_sq{i}.txt"
        if os.path.exists(sq):
            txt = open(sq).read().split("__amd_rocclr")[0].rstrip()
            pl += [f"SQ counter pass {i} (`tools/pmc_pass.sh`, `rocprofv3 --kernel-trace --pmc ...` on `tools/one_shape.py`: GPTQ 4096x4096, M=2048; "
                   "SQ_*_CYCLES of waves are in quad-cycles, summed over the chip):", "", "```", txt, "```", ""]
    open(f"{out}/{tag}_prefill_summary.md", "w").write("\n".join(pl) + "\n")
