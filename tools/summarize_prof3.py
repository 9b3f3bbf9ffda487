#!/usr/bin/env python3
"""Condense a tools/profile_round3.sh run directory (gpurun_out/prof_<tag>) into the small files committed under profiles/:
<tag>_bench_kernel_stats.csv, <tag>_bench_summary.md (per-launch table: duration, algorithmic bytes, PMC traffic),
<tag>_hqq_summary.md (BASELINE configs[3]: per-launch table + PMC traffic of the HQQ g64 batch-16 layers, 4 and 3 bits),
<tag>_prefill_kernel_stats.csv, <tag>_prefill_summary.md (per-shape TFLOP/s, SQ counters of the prefill GEMM)."""
import collections
import csv
import glob
import json
import os
import re
import sys

DECODE_RE = re.compile(r"qllm::strip\d?_kernel")  # strip_kernel and the M = 1 strip1_kernel

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/prof_{tag}"
out = "profiles"
os.makedirs(out, exist_ok=True)
H, I, G = 4096, 11008, 128


def alg(K, N):  # packed words + scales + packed zeros + x in + y out (bench.alg_bytes, M = 1)
    return K * N // 2 + (K // G) * N * 2 + (K // G) * N // 2 + 2 * K + 2 * N


ROLES = [("q/k/v (one grouped launch)", 3 * alg(H, H)), ("o_proj", alg(H, H)), ("gate/up (one grouped launch)", 2 * alg(H, I)),
         ("down_proj", alg(I, H))]


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


rows = list(csv.DictReader(open(glob.glob(f"{src}/trace/*kernel_stats.csv")[0])))
with open(f"{out}/{tag}_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        if "qllm::" in r["Name"]:
            w.writerow(r)

# dispatches of the decode kernel in issue order: every step is 32 x (q/k/v, o, gate/up, down)
tr = [r for r in csv.DictReader(open(glob.glob(f"{src}/trace/*kernel_trace.csv")[0])) if DECODE_RE.search(r["Kernel_Name"])]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
per_role = collections.defaultdict(list)
inst = {}
for i, r in enumerate(tr):
    role = i % 4
    per_role[role].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    inst[role] = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
pmc = {}
for kind in ("fetch", "write"):
    fs = glob.glob(f"{src}/pmc_{kind}/*counter_collection.csv")
    if not fs:
        continue
    rs = [r for r in csv.DictReader(open(fs[0])) if DECODE_RE.search(r["Kernel_Name"])]
    rs.sort(key=lambda r: int(r["Dispatch_Id"]))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for i, r in enumerate(rs):
        agg[i % 4][0] += 1
        agg[i % 4][1] += float(r["Counter_Value"])
    pmc[kind] = {k: v[1] / v[0] for k, v in agg.items()}

bench = json.loads(open(f"{src}/bench_under_rocprof.json").read().strip().splitlines()[-1])
full = None
if os.path.exists(f"gpurun_out/{tag}_bench.json"):
    try:
        full = json.loads(open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1])
        json.dump(full, open(f"{out}/{tag}_bench.json", "w"), indent=1)
    except Exception:  # noqa: BLE001
        full = None
L = [f"# {tag}: decode step under rocprofv3, per launch", "",
     "Command (on the GPU box, `cd /tmp && export TMPDIR=/tmp` first):",
     "`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 3 --no-extra`;",
     "HBM counters from two separate passes `--kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (5 steps) of the same command",
     "(`tools/profile_round3.sh`).  The step: 4 launches per decoder layer on one stream (q/k/v and gate/up grouped), the modules",
     "decoding from their native strip-major copies, every launch on the batch-1 kernel (`csrc/strip1_kernel.hpp`).", "",
     f"bench line under the profiler: value={bench['value']} {bench['unit']}, ms_per_step={bench['ms_per_step']}, "
     f"avg launch {bench['roofline']['avg_launch_us']} us incl. gaps", ""]
if full:
    L += [f"un-profiled default run of the same commit (`{tag}_bench.json`): value={full['value']} {full['unit']}, "
          f"ms_per_step={full['ms_per_step']}, roofline.frac={full['roofline']['frac']}, roofline.traffic={full['roofline']['traffic']} "
          f"bytes per launch (algorithmic {full['roofline']['bytes_per_launch']})", ""]
L += ["| launch | kernel instance | grid threads | block | dispatches | min us | median us | algorithmic MB | TB/s (algorithmic / median) | "
      "HBM traffic MB (2 x FETCH_SIZE + WRITE_SIZE) | traffic / algorithmic |", "|---|---|---|---|---|---|---|---|---|---|---|"]
tot_alg = tot_traffic = tot_med = 0.0
per_launch = {}
for role in range(4):
    v = sorted(per_role[role])
    if not v:
        continue
    name, a = ROLES[role]
    med = v[len(v) // 2] / 1e3
    f, wv = pmc.get("fetch", {}).get(role), pmc.get("write", {}).get(role)
    traffic = (2 * f + wv) * 1024 if f is not None and wv is not None else None
    k = inst[role]
    L.append(f"| {name} | `{k[0]}` | {k[1]} | {k[2]} | {len(v)} | {v[0] / 1e3:.2f} | {med:.2f} | {a / 1e6:.2f} | {a / med / 1e6:.2f} | " +
             (f"{traffic / 1e6:.1f} | {traffic / a:.3f} |" if traffic else "- | - |"))
    tot_alg += a
    tot_med += med
    if traffic:
        tot_traffic += traffic
        per_launch[name.split(" ")[0]] = int(traffic)
L += ["", f"Sum of the four medians: {tot_med:.2f} us per decoder layer = {32 * tot_med / 1e3:.3f} ms per token if nothing overlapped; "
      f"algorithmic bytes per layer {tot_alg / 1e6:.1f} MB.",
      "Template arguments of `strip1_kernel` (csrc/strip1_kernel.hpp, the M = 1 4-bit g128 decode kernel): <waves per block, k-steps "
      "per wave, exact-fit K (no tail predicate), staged x chunks per lane, ablation level (4 = full), timeline diagnostics, fused "
      "all-reduce epilogue>; of `strip_kernel`: <waves per block, strips (columns) per lane, k-steps per round, k-steps per group, "
      "staged x chunks per lane, bits, register-A, bf16, row tiles, strip-major layout, timeline diagnostics, one-round fold>.",
      "FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM section): doubled; unit KB."]
notes = f"{out}/{tag}_bench_notes.md"
if os.path.exists(notes):  # hand-written remarks kept next to the generated table
    L += [l.rstrip("\n") for l in open(notes)]
open(f"{out}/{tag}_bench_summary.md", "w").write("\n".join(L) + "\n")
print("\n".join(L))
if tot_traffic:
    json.dump({"source": f"profiles/{tag}_bench_summary.md (rocprofv3 PMC passes)",
               "bytes_per_launch_avg_over_step": int(tot_traffic / 4), "per_launch": per_launch,
               "algorithmic_bytes_per_launch_avg": int(tot_alg / 4)}, open(f"{out}/{tag}_pmc.json", "w"), indent=2)

# ---- BASELINE configs[3]: HQQ g64, batch 16 ---------------------------------------------------------------------------------
ht = glob.glob(f"{src}/hqq_trace/*kernel_trace.csv")
if ht:
    def hqq_alg(K, N, bits, M=16, g=64):  # packed words + fp16 scales + fp16 zero points + x + y
        return K * N * bits // 8 + 2 * (K // g) * N * 2 + 2 * M * K + 2 * M * N
    def bits_of(name):  # strip_kernel<NW, CPL, MAXS, SPG, XL, BITS, ...> / strip_dma_kernel<NW, CPL, SPG, BITS, BF16, MT>
        m = re.search(r"strip_dma_kernel<([^>]*)>", name)
        if m:
            return int(m.group(1).split(",")[3])
        if "panel_kernel<" in name:  # csrc/panel.hip: 4 bits only
            return 4
        if "strip1_kernel<" in name:  # csrc/strip1_kernel.hpp: 4 bits only
            return 4
        m = re.search(r"strip_kernel<([^>]*)>", name)
        return int(m.group(1).split(",")[5]) if m else 0
    def ours(name):
        return bool(DECODE_RE.search(name)) or "qllm::strip_dma_kernel" in name or "panel_kernel<" in name
    names = ["q/k/v (one grouped launch)", "o_proj", "gate/up (one grouped launch)", "down_proj"]
    shapes = [(H, 3 * H), (H, H), (H, 2 * I), (I, H)]
    tr = [r for r in csv.DictReader(open(ht[0])) if ours(r["Kernel_Name"])]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    Q = [f"# {tag}: BASELINE configs[3] -- HQQ g64 fp16 zero points, batch 16, per launch", "",
         "`rocprofv3 --kernel-trace --stats -- python tools/hqq_leg.py 10` (a 32-layer stack of each width through the modules: sibling",
         "groups, native layout; graph replay), HBM counters from `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of `tools/hqq_leg.py 3 8` (8 layers).",
         "Un-profiled lines of the same script:", "", "```"]
    Q += [l.rstrip() for l in open(f"{src}/hqq_leg.log") if l.startswith("hqq")]
    Q += ["```", ""]
    for bits in (4, 3):
        rows_b = [r for r in tr if bits_of(r["Kernel_Name"]) == bits]
        per = collections.defaultdict(list)
        inst_b = {}
        for i, r in enumerate(rows_b):
            per[i % 4].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            inst_b[i % 4] = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
        pm = {}
        for kind in ("fetch", "write"):
            fs = glob.glob(f"{src}/hqq_{kind}/*counter_collection.csv")
            if fs:
                rs = [r for r in csv.DictReader(open(fs[0])) if ours(r["Kernel_Name"]) and bits_of(r["Kernel_Name"]) == bits]
                rs.sort(key=lambda r: int(r["Dispatch_Id"]))
                agg = collections.defaultdict(lambda: [0, 0.0])
                for i, r in enumerate(rs):
                    agg[i % 4][0] += 1
                    agg[i % 4][1] += float(r["Counter_Value"])
                pm[kind] = {k: v[1] / v[0] for k, v in agg.items()}
        Q += [f"## {bits}-bit layers", "",
              "| launch | kernel instance | grid threads | block | dispatches | min us | median us | algorithmic MB | TB/s (algorithmic / median) | "
              "HBM traffic MB (2 x FETCH_SIZE + WRITE_SIZE) | traffic / algorithmic |", "|---|---|---|---|---|---|---|---|---|---|---|"]
        tm = ta = 0.0
        for role in range(4):
            v = sorted(per[role])
            if not v:
                continue
            K, N = shapes[role]
            a = hqq_alg(K, N, bits)
            med = v[len(v) // 2] / 1e3
            f, wv = pm.get("fetch", {}).get(role), pm.get("write", {}).get(role)
            traffic = (2 * f + wv) * 1024 if f is not None and wv is not None else None
            k = inst_b[role]
            Q.append(f"| {names[role]} | `{k[0]}` | {k[1]} | {k[2]} | {len(v)} | {v[0] / 1e3:.2f} | {med:.2f} | {a / 1e6:.2f} | {a / med / 1e6:.2f} | " +
                     (f"{traffic / 1e6:.1f} | {traffic / a:.3f} |" if traffic else "- | - |"))
            tm += med
            ta += a
        Q += ["", f"Sum of the four medians: {tm:.2f} us per decoder layer; algorithmic bytes per layer {ta / 1e6:.1f} MB = "
              f"{ta / tm / 1e6:.2f} TB/s = {ta / tm / 1e6 / 8.0:.3f} of 8 TB/s.",
              "Template arguments of `strip_dma_kernel`: <waves per block, strips per block, k-steps per group, bits, bf16 activations, row tiles, fp16-zero-point form>; "
              "of `panel_kernel` (csrc/panel.hip; from 17 rows): <row tiles, strips per wave, K halves per block, k-steps per group, bf16 activations, fp16-zero-point form>.", ""]
    notes = f"{out}/{tag}_hqq_notes.md"
    if os.path.exists(notes):
        Q += [l.rstrip("\n") for l in open(notes)]
    open(f"{out}/{tag}_hqq_summary.md", "w").write("\n".join(Q) + "\n")
    print("\n".join(Q))

# ---- prefill -------------------------------------------------------------------------------------------------------------
pf = glob.glob(f"{src}/prefill/*kernel_stats.csv")
if pf:
    rows = [r for r in csv.DictReader(open(pf[0])) if "qllm::" in r["Name"]]
    with open(f"{out}/{tag}_prefill_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    P = [f"# {tag}: prefill kernels", "",
         "`rocprofv3 --kernel-trace --stats --output-format csv -- python tools/kbench.py --m 2048 --iters 40 --layouts GPTQ GEMM`",
         f"(M = 2048, Llama-2-7B shapes; per-kernel stats in `{tag}_prefill_kernel_stats.csv`).  Per-shape throughput printed by the same",
         "run (graph replay, HIP events; GEMM = the AWQ layout):", "", "```"]
    P += [l.rstrip() for l in open(f"{src}/prefill_kbench.log") if l.startswith(("GPTQ", "GEMM"))]
    P += ["```", ""]
    tr2 = glob.glob(f"{src}/prefill/*kernel_trace.csv")
    if tr2:
        dd = collections.defaultdict(list)
        for r in csv.DictReader(open(tr2[0])):
            if "qllm::gemm" in r["Kernel_Name"]:
                dd[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        P += ["| kernel | grid threads | block | dispatches | min us | median us |", "|---|---|---|---|---|---|"]
        for k, v in sorted(dd.items()):
            v.sort()
            P.append(f"| `{k[0]}` | {k[1]} | {k[2]} | {len(v)} | {v[0] / 1e3:.1f} | {v[len(v) // 2] / 1e3:.1f} |")
        P.append("")
    # (round 6) the module step, per launch: dispatches of gemm3 in issue order cycle through q/k/v (grouped), o, gate/up (grouped), down
    ps = glob.glob(f"{src}/prefill_step/*kernel_trace.csv")
    if ps:
        M = 2048
        roles = [("q/k/v (one grouped launch)", 3 * 2.0 * M * H * H), ("o_proj", 2.0 * M * H * H), ("gate/up (one grouped launch)", 2 * 2.0 * M * H * I),
                 ("down_proj", 2.0 * M * I * H)]
        tr3 = [r for r in csv.DictReader(open(ps[0])) if "qllm::gemm3_kernel" in r["Kernel_Name"]]
        tr3.sort(key=lambda r: int(r["Start_Timestamp"]))
        if len(tr3) % 4 == 0 and tr3:
            per3 = collections.defaultdict(list)
            g3 = {}
            for i, r in enumerate(tr3):
                per3[i % 4].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                g3[i % 4] = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
            P += ["The prefill step as the modules run it (`rocprofv3 --kernel-trace --stats -- python tools/prefill_legs.py 10 awq`: 4 decoder layers,",
                  "M = 2048, AWQ w4 g128 native layout, fp16; 4 launches of `gemm3_kernel` per layer since round 6), per launch:", "",
                  "| launch | blocks | dispatches | min us | median us | TFLOP/s (median) | of 2.5 PF |", "|---|---|---|---|---|---|---|"]
            tot = 0.0
            for role in range(4):
                v = sorted(per3[role])
                med = v[len(v) // 2] / 1e3
                tot += med
                P.append(f"| {roles[role][0]} | {g3[role]} | {len(v)} | {v[0] / 1e3:.1f} | {med:.1f} | {roles[role][1] / med / 1e6:.0f} | {roles[role][1] / med / 1e6 / 2500:.3f} |")
            fl = sum(r[1] for r in roles)
            P += ["", f"Sum of the four medians: {tot:.1f} us per decoder layer = {fl / tot / 1e6:.0f} TFLOP/s = {fl / tot / 1e6 / 2500:.3f} of 2.5 PF (un-profiled line of the same script: "
                  + "; ".join(l.strip() for l in open(f"{src}/prefill_step.log") if l.startswith("awq")) + ").", ""]
    for nm in ("sq1",):
        fn = f"gpurun_out/pmc_{tag}_{nm}.txt"
        if os.path.exists(fn):
            P += [f"SQ counters, pass {nm} (`tools/pmc_pass.sh`, `python tools/one_shape.py`: GPTQ 4096x4096, M = 2048, per launch):", "", "```"]
            P += [l.rstrip() for l in open(fn)][:40]
            P += ["```", ""]
    fn = f"gpurun_out/{tag}_hipblaslt_ref.log"
    if os.path.exists(fn):
        P += ["Context (`python tools/one_shape.py --ref`): the same shape as a dense fp16 `torch.matmul` (hipBLASLt), weights already dequantised:", "", "```"]
        P += [l.rstrip() for l in open(fn) if "TFLOP" in l]
        P += ["```", ""]
    notes = f"{out}/{tag}_prefill_notes.md"
    if os.path.exists(notes):  # hand-written reading of the numbers above, kept next to them
        P += [l.rstrip("\n") for l in open(notes)]
    open(f"{out}/{tag}_prefill_summary.md", "w").write("\n".join(P) + "\n")
    print("\n".join(P))
