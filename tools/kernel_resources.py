#!/usr/bin/env python3
"""Static resource table of every kernel in libqllm_mi355x (hipcc -S for gfx950, code-object metadata): VGPRs, spills, SGPRs,
static LDS, and the waves per SIMD the register allocation allows (512-entry file, granule 8: MI355X_MICROARCH.md).
Usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.md   (no GPU needed)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qllm_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("(anonymous namespace)::", "").replace("qllm::", "").split("(")[0].replace("void ", "") for o in out]


rows = []
for src in ("strip1.hip", "strip_sm.hip", "strip_sm_ra.hip", "strip_dma_g32.hip", "strip_dma_g64.hip", "strip_dma_g128.hip", "strip.hip", "native.hip", "skinny.hip", "gemm3.hip", "panel.hip", "gemm2.hip", "gemm.hip", "dequant.hip", "gather.hip", "comm.hip", "ortblob.hip"):
    asm = f"/tmp/qllm_kres_{src}.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                    os.path.join(CSRC, src), "-o", asm], check=True, capture_output=True)
    for block in open(asm).read().split("\n  - ")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", block)  # noqa: E731
        if not g("name") or not g("vgpr_count"):
            continue
        vg, sp = int(g("vgpr_count").group(1)), int(g("vgpr_spill_count").group(1))
        alloc = (vg + 7) // 8 * 8
        rows.append((src, g("name").group(1), vg, sp, int(g("sgpr_count").group(1)), int(g("group_segment_fixed_size").group(1)),
                     int(g("max_flat_workgroup_size").group(1)), min(8, 512 // max(alloc, 8))))
names = demangle([r[1] for r in rows])
print("# Static kernel resources (hipcc -O3 --offload-arch=gfx950, code-object metadata; `python tools/kernel_resources.py`)\n")
print("Dynamic LDS (strip: <= 156 KB, gemm2: 128 KB, gemm: 64-160 KB, skinny: per plan) is not in the static column.\n")
print("| file | kernel | VGPRs | spilled | SGPRs | static LDS B | max block | waves/SIMD by registers |")
print("|---|---|---|---|---|---|---|---|")
for (src, _, vg, sp, sg, lds, wg, occ), n in sorted(zip(rows, names), key=lambda t: (t[0][0], t[1])):
    print(f"| {src} | `{n}` | {vg} | {sp} | {sg} | {lds} | {wg} | {occ} |")
