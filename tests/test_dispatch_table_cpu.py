"""The kernel-selection table, checked without a GPU through the pure-host entry point qllm_plan_describe (descriptors carry
fake, aligned, never-dereferenced pointers).  Guards against silent fallbacks: earlier in this round an over-estimated LDS
size quietly sent every M=16 launch to the slower split-K kernel."""
import ctypes as C

import pytest

from qllm_amd import _lib

GPTQ, AWQ, HQQ, NATIVE, NATIVE_F16Z = _lib.LAYOUT_GPTQ, _lib.LAYOUT_AWQ_GEMM, _lib.LAYOUT_HQQ, _lib.LAYOUT_NATIVE, _lib.LAYOUT_NATIVE_F16Z


@pytest.fixture(scope="module")
def lib():
    if not _lib.is_built():
        pytest.skip("libqllm_mi355x.so not built")
    return _lib.load()


def W(K, N, g=128, bits=4, layout=GPTQ, zeros=16, g_idx=None):
    return _lib.QllmWeight(16, 16, zeros, g_idx, None, K, N, g, bits, layout, 0)


def plan(lib, ws, m, have_ws=1):
    arr = (_lib.QllmWeight * len(ws))(*ws)
    buf = C.create_string_buffer(256)
    assert lib.qllm_plan_describe(arr, len(ws), m, have_ws, buf, 256) == 0, _lib.last_error()
    return buf.value.decode()


def test_llama7b_decode_routes(lib):
    attn, up, down = W(4096, 4096), W(4096, 11008), W(11008, 4096)
    # batch 1: every linear of the stack on the full-K strip kernel, LDS-slab form
    assert plan(lib, [attn], 1) == "strip nw=16 cpl=1 spw=8 form=lds-slab row_tiles=1"
    assert plan(lib, [down], 1) == "strip nw=16 cpl=1 spw=24 form=lds-slab row_tiles=1"          # one round of 24 loads
    assert plan(lib, [attn] * 3, 1) == "strip nw=8 cpl=4 spw=16 form=lds-slab row_tiles=1"      # grouped q/k/v: 64-column strips
    assert plan(lib, [up] * 2, 1) == "strip nw=8 cpl=4 spw=16 form=lds-slab row_tiles=1"        # grouped gate/up
    # batch 5..16: register-A form; 17..32: two row tiles; 33..64: four row tiles on the 4096x4096 shape, split-K kernel elsewhere
    for m in (5, 8, 16):
        assert "form=register-A row_tiles=1" in plan(lib, [attn], m)
        assert "form=register-A row_tiles=1" in plan(lib, [down], m)
        assert plan(lib, [attn] * 3, m).startswith("strip nw=8 cpl=4")
    assert "form=lds-slab" in plan(lib, [attn], 4)
    for m in (17, 32):
        assert plan(lib, [attn], m) == "strip nw=8 cpl=1 spw=16 form=register-A row_tiles=2"
        assert plan(lib, [attn] * 3, m).endswith("row_tiles=2")
    for m in (33, 64):
        assert plan(lib, [attn], m) == "strip nw=8 cpl=1 spw=16 form=register-A row_tiles=4"   # measured 1.5x over split-K
        assert plan(lib, [up], m) == "gemm2 tile=256x128 split_k=2"       # the wide shapes: the 256-row-tile GEMM (round 3:
        assert plan(lib, [down], m) == "gemm2 tile=256x128 split_k=8"     #   40 us against 48-53 split-K and 42-59 strips)
        assert plan(lib, [attn] * 3, m).startswith("skinny tile_cols=64")  # grouped launches have no GEMM form
        assert plan(lib, [W(4096, 11000)], m).startswith("skinny")          # N % 128 != 0: split-K decode kernel


def test_llama7b_prefill_routes(lib):
    attn, up, down = W(4096, 4096), W(4096, 11008), W(11008, 4096)
    g3 = "gemm3 tile=256x128 matrix-waves=8 staging-waves=4"
    assert plan(lib, [attn], 2048) == g3                                   # 256 tiles, one per CU: the wave-specialised kernel
    assert plan(lib, [attn], 8192) == g3
    assert plan(lib, [down], 2048) == g3 and plan(lib, [up], 2048) == g3
    assert plan(lib, [W(4096 + 64, 4096, 64)], 2048) == g3                  # an odd number of k-tiles is fine (tail barrier)
    assert plan(lib, [attn], 512) == "gemm2 tile=256x128 split_k=4"       # 64 tiles -> 4 blocks per tile
    assert plan(lib, [attn], 256) == "gemm2 tile=256x128 split_k=8"
    assert plan(lib, [down], 1024) == g3 + " split_k=2"                    # the wave-specialised kernel also splits K
    assert plan(lib, [attn], 1024) == g3 + " split_k=2"                    # (measured 47.7 vs 52.2 us, 100.6 vs 110.2 us for `down`)
    # round 3 (profiles/r03_mid_m.md): the wave-specialised kernel from M = 384 on the 11008-wide shapes, from 768 on 4096 x 4096
    assert plan(lib, [down], 512) == g3 + " split_k=4" and plan(lib, [down], 384) == g3 + " split_k=4"
    assert plan(lib, [up], 512) == g3 and plan(lib, [up], 383).startswith("gemm2")
    assert plan(lib, [attn], 768) == g3 + " split_k=2" and plan(lib, [attn], 767).startswith("gemm2")
    assert plan(lib, [attn], 512, have_ws=0) == "gemm2 tile=256x128 split_k=1"  # no workspace: no split, still fused
    assert plan(lib, [attn], 128) == "gemm2 tile=256x128 split_k=8"       # 64 < M < 192: k-loop-bound, split-K (2.5-6x the 128x128 kernel)
    assert plan(lib, [attn], 65) == "gemm2 tile=256x128 split_k=8"
    assert plan(lib, [W(4096, 4000)], 2048) == "gemm tile=128x128"        # ragged N
    assert plan(lib, [W(4096, 4096, layout=AWQ)], 2048) == g3               # AWQ layout read in place
    assert plan(lib, [W(4096, 4096, layout=AWQ)], 512) == "gemm2 tile=256x128 split_k=4"
    assert plan(lib, [W(4096, 11008, layout=AWQ)], 48) == "gemm2 tile=256x128 split_k=2"   # 33..64 rows, AWQ in place


def test_other_layouts_and_widths(lib):
    # AWQ in place at decode sizes: split-K kernel (the modules hand the strip kernel their row-stream view instead)
    assert plan(lib, [W(4096, 4096, layout=AWQ)], 1).startswith("skinny tile_cols=128")
    # HQQ g64: batch 16 on the register-A strips; long K at batch 1 too (the 24-load slab variant would spill)
    assert "form=register-A" in plan(lib, [W(11008, 4096, 64, 4, HQQ)], 1)
    assert "form=lds-slab" in plan(lib, [W(4096, 4096, 64, 4, HQQ)], 1)
    assert "form=register-A" in plan(lib, [W(4096, 4096, 64, 4, HQQ)], 16)
    # 3 bits: fused for HQQ / symmetric zeros at decode sizes only
    assert plan(lib, [W(4096, 4096, 64, 3, HQQ)], 16).startswith("strip nw=16 cpl=1")
    assert plan(lib, [W(4096, 4096, 128, 3, GPTQ, zeros=None)], 1).startswith("strip")
    assert plan(lib, [W(4096, 4096, 128, 3, GPTQ)], 1).startswith("strip")          # packed 3-bit zeros: funnel shift over two words
    assert plan(lib, [W(4096, 4096, 64, 3, HQQ)], 300) == "gemm3 tile=256x128 matrix-waves=8 staging-waves=4 bits=3 split_k=4"
    assert plan(lib, [W(4096, 4096, 64, 3, HQQ)], 300, have_ws=0).endswith("bits=3")   # no workspace: unsplit, still fused
    assert plan(lib, [W(4096, 4096, 64, 3, HQQ)], 2048) == "gemm3 tile=256x128 matrix-waves=8 staging-waves=4 bits=3"
    assert plan(lib, [W(4096, 11008, 128, 3, GPTQ)], 1024).endswith("bits=3")
    assert plan(lib, [W(4096, 4000, 64, 3, HQQ)], 300).startswith("unsupported")    # ragged N: dequant + GEMM
    # 2 / 5 / 6 / 7 / 8 bits at decode sizes: the bit-stream matvec (round 6; until then dequant + GEMM); prefill sizes stay "unsupported"
    assert plan(lib, [W(4096, 4096, 128, 8)], 1) == "bitgemv bits=8 cols=32 waves=8 split_k=4"      # 128 column blocks x 4 K parts: two per CU
    assert plan(lib, [W(4096, 4096, 64, 2, HQQ)], 16) == "bitgemv bits=2 cols=32 waves=8 split_k=4"
    assert plan(lib, [W(4096, 11008, 128, 5)], 4) == "bitgemv bits=5 cols=32 waves=8 split_k=2"
    assert plan(lib, [W(4096, 1024, 128, 8)], 1) == "bitgemv bits=8 cols=32 waves=8 split_k=8"      # narrow layer: 32 column blocks, one unit per lane slot
    assert plan(lib, [W(4096, 1024, 128, 8)], 1, have_ws=0).endswith("split_k=1")
    assert plan(lib, [W(4096, 4096, 128, 8)], 17).startswith("unsupported") and plan(lib, [W(4096, 4096, 128, 6, g_idx=16)], 1).startswith("unsupported")
    assert plan(lib, [W(4096, 4096, 128, 8)] * 2, 1).startswith("unsupported")     # (no grouped form: the layers run one by one)
    # raw act-order descriptors (the modules use a row-sorted view instead): in-place gather in the 128x128 kernel
    assert plan(lib, [W(4096, 4096, g_idx=16)], 300) == "gemm tile=128x128 act-order-gather"
    # narrow layers: the full-K strips even when they cannot fill the chip (measured 2x faster than split-K's three round trips)
    assert plan(lib, [W(4096, 1024)], 1).startswith("strip nw=16 cpl=1")
    assert plan(lib, [W(4096, 1024)] * 3, 1).startswith("strip")                    # grouped: 192 strips together
    assert plan(lib, [W(4096, 64)], 1).startswith("skinny")                         # 4 strips: below the minimum of 8
    # Llama-2-70B and its 8-way tensor-parallel shards (BASELINE configs[4]): 32-column strips where they alone give ~1 block per CU
    assert plan(lib, [W(8192, 1024), W(8192, 128), W(8192, 128)], 1).startswith("strip nw=16 cpl=1")   # q/k/v shard: 80 strips
    assert plan(lib, [W(8192, 3584)] * 2, 1).startswith("strip nw=16 cpl=2")                            # gate/up shard: 224 blocks
    assert plan(lib, [W(8192, 8192)], 1).startswith("strip nw=16 cpl=2")
    assert plan(lib, [W(3584, 8192)], 1).startswith("strip nw=16 cpl=2")                                # row-parallel down shard
    assert plan(lib, [W(8192, 28672)], 1).startswith("strip nw=8 cpl=4")
    assert plan(lib, [W(8192, 3584)] * 2, 16).startswith("strip nw=16 cpl=1")                           # M > 4: measured forms only
    # group sizes the strip kernel does not serve
    assert plan(lib, [W(4096, 4096, 32)], 1).startswith("skinny")


def test_plan_describe_validates(lib):
    arr = (_lib.QllmWeight * 1)(W(4096, 4096, bits=9))
    buf = C.create_string_buffer(256)
    assert lib.qllm_plan_describe(arr, 1, 1, 1, buf, 256) == _lib.QLLM_ERR_INVALID and "bits" in _lib.last_error()
    assert lib.qllm_plan_describe(arr, 0, 1, 1, buf, 256) == _lib.QLLM_ERR_INVALID
    assert lib.qllm_plan_describe(None, 1, 1, 1, buf, 256) == _lib.QLLM_ERR_INVALID


# ---- the native (strip-major) layout: plans, validation, conversions -- pure host code ---------------------------------------
def test_native_layout_decode_routes(lib):
    """Every decode launch of the Llama-2-7B stack on the strip-major kernel: 16-column strips, all of a launch's weights in ONE
    round per wave (8 waves x 16 k-steps at K = 4096, 16 waves x 24 at K = 11008)."""
    sm = " layout=strip-major"
    attn, up, down = W(4096, 4096, layout=NATIVE), W(4096, 11008, layout=NATIVE), W(11008, 4096, layout=NATIVE)
    # batch 1, 4 bits, 128-wide groups: the specialised kernel (strip1_kernel.hpp, round 5): the layer of a grouped launch is blockIdx.y
    assert plan(lib, [attn], 1) == "strip1 nw=4 round=32 exact grid=strips x 1" + sm     # one block per CU: four waves, one per SIMD
    assert plan(lib, [attn] * 3, 1) == "strip1 nw=8 round=16 exact grid=strips x 3" + sm
    assert plan(lib, [up] * 2, 1) == "strip1 nw=8 round=16 exact grid=strips x 2" + sm
    assert plan(lib, [down], 1) == "strip1 nw=15 round=24 grid=strips x 1" + sm      # (344 k-steps: no sixteenth, dead wave)
    # ... the Llama-2-70B TP = 8 shards (BASELINE configs[4]): q/k/v (ragged widths), o (K = 1024), gate/up, down (K = 3584)
    assert plan(lib, [W(8192, 1024, layout=NATIVE), W(8192, 128, layout=NATIVE), W(8192, 128, layout=NATIVE)], 1) == "strip1 nw=8 round=32 exact grid=strips x 3" + sm
    assert plan(lib, [W(1024, 8192, layout=NATIVE)], 1) == "strip1 nw=4 round=8 exact grid=strips x 1" + sm
    assert plan(lib, [W(3584, 8192, layout=NATIVE)], 1) == "strip1 nw=7 round=16 exact grid=strips x 1" + sm
    # ... other group sizes, 3 bits and K beyond 512 k-steps stay on the general strip kernel
    assert plan(lib, [W(4096, 4096, 64, layout=NATIVE)], 1) == "strip1 nw=4 round=32 exact g64 grid=strips x 1" + sm      # (round 6: 64-wide groups on the batch-1 kernel)
    assert plan(lib, [W(4096, 4096, 32, layout=NATIVE)], 1).startswith("strip nw=")                                       # 32-wide groups: the general kernel
    assert plan(lib, [W(4096, 4096, 128, 3, NATIVE)], 1) == "strip1 nw=4 round=32 exact bits=3 grid=strips x 1" + sm          # (round 6: 3-bit layers too)
    assert plan(lib, [W(28672, 4096, 128, 3, NATIVE)], 1).startswith("strip nw=16")                                           # 3 bits: K <= 16384
    assert plan(lib, [W(28672, 8192, layout=NATIVE)], 1) == "strip1 nw=16 round=56 exact grid=strips x 1" + sm     # (round 6; the general kernel until then)
    assert plan(lib, [W(36864, 8192, layout=NATIVE)], 1).startswith("strip nw=16")                                # K > 32768: the general strip kernel
    assert plan(lib, [W(18944, 3584, layout=NATIVE)], 1) == "strip1 nw=16 round=40 grid=strips x 1" + sm       # round 6: K up to 24576 (Qwen2-7B down_proj)
    assert plan(lib, [W(24576, 4096, layout=NATIVE)], 1) == "strip1 nw=16 round=48 exact grid=strips x 1" + sm
    assert plan(lib, [W(32768, 4096, layout=NATIVE)], 1) == "strip1 nw=16 round=64 exact grid=strips x 1" + sm
    # M = 2..32: strip_dma.hpp (activations through LDS by DMA); one strip per 16-wave block while the strips fit one round of CUs
    # (round 6: batches 2..4 on 128-wide groups ride the batch-1 kernel's four-row forms -- 32.0 / 33.6 / 35.0 us per 7B layer against
    #  37.6 / 38.0 / 38.5 on strip_dma; QLLM_STRIP1_MAX_M = 1 restores strip_dma)
    for m in (2, 3, 4):
        assert plan(lib, [attn], m) == "strip1 nw=4 round=32 exact rows=4 grid=strips x 1" + sm
        assert plan(lib, [attn] * 3, m) == "strip1 nw=8 round=16 exact rows=4 grid=strips x 3" + sm
        assert plan(lib, [down], m) == "strip1 nw=15 round=24 rows=4 grid=strips x 1" + sm
        assert plan(lib, [W(4096, 4096, 64, layout=NATIVE)], m).startswith("strip nw=16 cpl=1 spw=8 form=dma-A")     # 64-wide groups: strip_dma
        assert plan(lib, [W(18944, 3584, layout=NATIVE)], m).startswith("strip nw=16 cpl=1 spw=40 form=dma-A")       # K > 16384: strip_dma
    from qllm_amd import ops as _ops
    try:
        _ops.set_knob("QLLM_STRIP1_MAX_M", 1)
        assert plan(lib, [attn], 2) == "strip nw=16 cpl=1 spw=8 form=dma-A row_tiles=1" + sm
    finally:
        _ops.reset_knobs()
    for m in (5, 16):
        assert plan(lib, [attn], m) == "strip nw=16 cpl=1 spw=8 form=dma-A row_tiles=1" + sm
        # K >= 2 N: round 4 sent 9..16 rows to the panel kernel; with the strips' scale / zero tables in LDS (round 5) the one-strip
        # blocks win again (M = 16: 15.0 -> 13.5-14.4 us at g64, 13.5 -> 13.1 at g128; profiles/r05_batch16.md)
        assert plan(lib, [down], m) == "strip nw=16 cpl=1 spw=24 form=dma-A row_tiles=1" + sm
    assert plan(lib, [W(11008, 4096, 64, 4, NATIVE_F16Z)], 16) == "strip nw=16 cpl=1 spw=22 form=dma-A row_tiles=1" + sm   # (BASELINE configs[3] down_proj)
    assert plan(lib, [W(11008, 4096, 64, 3, NATIVE_F16Z)], 16) == "strip nw=16 cpl=1 spw=22 form=dma-A row_tiles=1" + sm   # (3 bits too)
    assert plan(lib, [W(11008, 4096, 64, 3, NATIVE_F16Z)], 17) == "panel cols=64 row_tiles=2 k_halves=2 split_k=4 bits=3" + sm
    assert plan(lib, [W(11008, 4096, 64, 3, NATIVE_F16Z)], 8).startswith("strip ")
    # ... wide (grouped) launches: blocks of several adjacent strips share one activation stream -- as many as make the launch ONE
    # round of blocks: q/k/v 768 strips -> 192 blocks of four; gate/up 1376 strips -> 230 blocks of six, the last of each layer ragged
    assert plan(lib, [attn] * 3, 16) == "strip nw=8 cpl=4 spw=16 form=dma-A row_tiles=1" + sm
    assert plan(lib, [up] * 2, 8) == "strip nw=8 cpl=6 spw=16 form=dma-A row_tiles=1" + sm
    assert plan(lib, [up], 16) == "strip nw=8 cpl=3 spw=16 form=dma-A row_tiles=1" + sm    # (round 5: 688 strips -> 230 blocks of three)
    assert plan(lib, [W(4096, 11008, 64, 4, NATIVE_F16Z)] * 2, 16) == "strip nw=8 cpl=6 spw=16 form=dma-A row_tiles=1" + sm
    assert plan(lib, [W(4096, 4096, 64, 3, NATIVE_F16Z)] * 3, 16) == "strip nw=8 cpl=4 spw=16 form=dma-A row_tiles=1" + sm
    # 3 bits: six strips where the ring fits the registers (64-wide groups, fp16 zero points: HQQ) -- ONE round of 230 blocks (round 5); else four
    assert plan(lib, [W(4096, 11008, 64, 3, NATIVE_F16Z)] * 2, 16) == "strip nw=8 cpl=6 spw=16 form=dma-A row_tiles=1" + sm
    assert plan(lib, [W(4096, 11008, 128, 3, NATIVE)] * 2, 16) == "strip nw=8 cpl=3 spw=16 form=dma-A row_tiles=1" + sm   # (two rounds either way: fewer bytes per block)
    # single layers from 17 rows: the panel kernel -- except up to 4096 x 4096 at 17..32 rows, where the two-row-tile strips are
    # faster (o_proj 9.1-9.7 -> 7.7-8.9 us, round 5)
    assert plan(lib, [attn], 32) == plan(lib, [attn], 17) == "strip nw=8 cpl=1 spw=16 form=dma-A row_tiles=2" + sm
    assert plan(lib, [down], 32) == plan(lib, [down], 17) == "panel cols=64 row_tiles=2 k_halves=2 split_k=4" + sm
    assert plan(lib, [up], 24) == "panel cols=64 row_tiles=2 k_halves=2 split_k=1" + sm
    assert plan(lib, [attn] * 3, 32) == "panel cols=64 row_tiles=2 k_halves=2 split_k=1 layers=3" + sm
    assert plan(lib, [up] * 2, 16) == "strip nw=8 cpl=6 spw=16 form=dma-A row_tiles=1" + sm
    assert plan(lib, [up] * 2, 17) == "panel cols=64 row_tiles=2 k_halves=2 split_k=1 layers=2" + sm   # groups from 17 rows: ONE panel launch
    assert plan(lib, [attn] * 3, 17) == "panel cols=64 row_tiles=2 k_halves=2 split_k=1 layers=3" + sm
    assert plan(lib, [attn] * 3, 128) == "panel cols=64 row_tiles=8 k_halves=1 split_k=1 layers=3" + sm
    assert plan(lib, [W(4096, 1024, layout=NATIVE)] * 3, 64) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4 layers=3" + sm   # 48 panels x 4
    assert plan(lib, [attn] * 3, 129).startswith("unsupported")
    # Llama-2-70B shapes at batch 16: 512 strips -> 256 blocks of two; q/k/v (GQA) 640 strips -> 215 blocks of three; gate/up 3584
    # strips -> six per block; the TP = 8 shards of q/k/v (80 strips) stay one strip per 16-wave block
    assert plan(lib, [W(8192, 8192, layout=NATIVE)], 16) == "strip nw=8 cpl=2 spw=32 form=dma-A row_tiles=1" + sm
    assert plan(lib, [W(8192, 8192, layout=NATIVE), W(8192, 1024, layout=NATIVE), W(8192, 1024, layout=NATIVE)], 16) == "strip nw=8 cpl=3 spw=32 form=dma-A row_tiles=1" + sm
    assert plan(lib, [W(8192, 28672, layout=NATIVE)] * 2, 16) == "strip nw=8 cpl=6 spw=32 form=dma-A row_tiles=1" + sm
    assert plan(lib, [W(28672, 8192, layout=NATIVE)], 8) == "strip nw=8 cpl=2 spw=112 form=dma-A row_tiles=1" + sm
    assert plan(lib, [W(28672, 8192, layout=NATIVE)], 16) == "strip nw=8 cpl=2 spw=112 form=dma-A row_tiles=1" + sm
    assert plan(lib, [W(28672, 8192, layout=NATIVE)], 17) == "panel cols=64 row_tiles=2 k_halves=2 split_k=2" + sm
    assert plan(lib, [W(8192, 1024, layout=NATIVE), W(8192, 128, layout=NATIVE), W(8192, 128, layout=NATIVE)], 16) == "strip nw=16 cpl=1 spw=16 form=dma-A row_tiles=1" + sm
    # 33 <= M <= 128, single 4-bit layers: the panel kernel (panel.hip, round 4): 64-column panels, two K halves per block up to 64
    # rows, split over K until the panels cover the CUs (at least two K-tiles per part); grouped launches keep the four-row-tile strips
    assert plan(lib, [attn], 33) == plan(lib, [attn], 64) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4" + sm
    assert plan(lib, [attn], 65) == plan(lib, [attn], 128) == "panel cols=64 row_tiles=8 k_halves=1 split_k=4" + sm
    assert plan(lib, [attn], 64, have_ws=0) == "panel cols=64 row_tiles=4 k_halves=2 split_k=1" + sm      # no workspace: no split
    assert plan(lib, [attn], 129) == "gemm2 tile=256x128 split_k=8" + sm
    assert plan(lib, [W(4096, 1024, 128, 3, NATIVE)] * 3, 32) == "panel cols=64 row_tiles=2 k_halves=2 split_k=4 layers=3 bits=3" + sm
    assert plan(lib, [W(4096, 1024, 128, 3, NATIVE)] * 3, 16).startswith("strip ")                 # (3 bits below 17 rows: strips)
    assert plan(lib, [W(4096, 1024, 128, 3, NATIVE)] * 3, 65).startswith("unsupported")            # (3-bit panels stop at 64 rows)
    assert plan(lib, [W(4096, 4096, 64, layout=NATIVE_F16Z)], 48) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4" + sm
    assert plan(lib, [W(4096, 4032, layout=NATIVE)], 48) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4" + sm   # N % 64 == 0 is enough
    assert plan(lib, [W(4096, 4048, layout=NATIVE)], 48).startswith("strip ")                                        # ... N % 16 is not
    # strip_dma's byte offsets are 32-bit: a layer of 2 GB of packed words stays on the register-A form (64-bit pointers)
    assert "form=register-A" in plan(lib, [W(65536, 65536, layout=NATIVE)], 16) and "form=dma-A" in plan(lib, [W(65536, 32768, layout=NATIVE)], 8)
    assert plan(lib, [up], 64) == "panel cols=64 row_tiles=4 k_halves=2 split_k=1" + sm     # 172 panels: no cross-block sum at all
    assert plan(lib, [down], 33) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4" + sm   # 344 k-steps in 8 parts of 44
    assert plan(lib, [up], 128) == "gemm2 tile=256x128 split_k=2" + sm   # 65..128 rows: layers of up to 2^25 weights only (bf16 loses above)
    assert plan(lib, [down], 100) == "gemm2 tile=256x128 split_k=8" + sm
    assert plan(lib, [up] * 2, 128).startswith("unsupported")              # ... groups of up to 16384 columns
    # shard shapes of Llama-2-70B (TP = 8): short K -> 4-wave blocks, K = 8192 -> 8 waves x one round of 32
    assert plan(lib, [W(1024, 8192, layout=NATIVE)], 1).startswith("strip1 nw=4 round=8 exact")
    assert plan(lib, [W(8192, 1024, layout=NATIVE)], 1).startswith("strip1 nw=8 round=32 exact")
    assert plan(lib, [W(1024, 8192, 64, layout=NATIVE)], 1).startswith("strip1 nw=4 round=8 exact g64")           # (64-wide groups too since round 6)
    assert plan(lib, [W(8192, 1024, layout=NATIVE)], 3).startswith("strip1 nw=8 round=32 exact rows=4")
    assert plan(lib, [W(8192, 1024, layout=NATIVE)], 5).startswith("strip nw=16 cpl=1 spw=16 form=dma-A")
    assert plan(lib, [W(28672, 1024, layout=NATIVE)], 1).startswith("strip1 nw=16 round=56 exact")                # (round 6: one round of 56)
    assert plan(lib, [W(36864, 1024, layout=NATIVE)], 1).startswith("strip nw=16 cpl=1 spw=72 form=lds-slab")     # beyond 32768: three rounds of 24
    # g64 / 3 bits / fp16 zero points: slab form for short chunks at batch 1, register-A beyond (no spilling instantiation is built)
    h4, h3 = W(4096, 4096, 64, 4, NATIVE_F16Z), W(4096, 4096, 64, 3, NATIVE_F16Z)
    assert plan(lib, [h4], 1) == "strip1 nw=4 round=32 exact g64 grid=strips x 1" + sm   # (round 6: HQQ's 64-wide groups on the batch-1 kernel)
    assert plan(lib, [h4], 2).startswith("strip nw=16 cpl=1 spw=8 form=dma-A")          # 64-wide groups, 3 bits: strip_dma from two rows
    assert plan(lib, [h3], 4).startswith("strip nw=16 cpl=1 spw=8 form=dma-A")
    assert plan(lib, [W(11008, 4096, 64, 4, NATIVE_F16Z)], 1) == "strip1 nw=15 round=24 g64 grid=strips x 1" + sm   # (was the register-A form: 12.3 us)
    assert plan(lib, [W(28672, 4096, 64, 4, NATIVE_F16Z)], 1).startswith("strip nw=16")                             # 64-wide groups: K <= 24576
    assert plan(lib, [h3], 1) == "strip1 nw=4 round=32 exact g64 bits=3 grid=strips x 1" + sm   # (round 6; the lds-slab form until then: 43 -> 34 us per 7B layer)
    assert plan(lib, [h3], 16).startswith("strip nw=16 cpl=1 spw=8 form=dma-A")
    # 3 bits from 17 rows: the panel kernel (exact q - z from the slot-scaled patterns), up to 64 rows; the 256-row tiles above
    assert plan(lib, [h3], 32) == "strip nw=8 cpl=1 spw=16 form=dma-A row_tiles=2" + sm      # (up to 4096 x 4096 and 32 rows: strips)
    assert plan(lib, [W(4096, 11008, 64, 3, NATIVE_F16Z)], 32) == "panel cols=64 row_tiles=2 k_halves=2 split_k=1 bits=3" + sm
    assert plan(lib, [h3], 48) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4 bits=3" + sm
    assert plan(lib, [h3], 65).startswith("gemm3") and "bits=3" in plan(lib, [h3], 65)
    assert plan(lib, [W(4096, 4096, 128, 3, NATIVE)], 33) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4 bits=3" + sm   # packed 3-bit zero points
    # 32-wide groups (4 bits, native layout only): one-round lds-slab blocks at batch 1 when a wave's chunk is exactly 8 k-steps (else
    # the register-A form); M = 2..32 the DMA form with shorter rings, blocks of one or two strips, chunks rounded to whole k-step pairs
    g32 = lambda K, N: W(K, N, 32, layout=NATIVE)  # noqa: E731
    assert plan(lib, [g32(4096, 4096)], 1) == "strip nw=16 cpl=1 spw=8 form=lds-slab row_tiles=1" + sm
    assert plan(lib, [g32(4096, 4096)] * 3, 1) == "strip nw=16 cpl=1 spw=8 form=lds-slab row_tiles=1" + sm
    assert plan(lib, [g32(1024, 8192)], 1) == "strip nw=4 cpl=1 spw=8 form=lds-slab row_tiles=1" + sm
    assert plan(lib, [g32(11008, 4096)], 1) == "strip nw=16 cpl=1 spw=22 form=register-A row_tiles=1" + sm
    assert plan(lib, [g32(2048, 4096)], 1) == "strip nw=16 cpl=1 spw=4 form=register-A row_tiles=1" + sm
    for m in (2, 4, 8, 16):
        # (one strip per block, K <= 4096, M <= 8: register-A measured faster than the three-slot ring)
        assert plan(lib, [g32(4096, 4096)], m) == "strip nw=16 cpl=1 spw=8 form=%s row_tiles=1" % ("register-A" if m <= 8 else "dma-A") + sm
        assert plan(lib, [g32(4096, 11008)] * 2, m) == "strip nw=8 cpl=2 spw=16 form=dma-A row_tiles=1" + sm
        assert plan(lib, [g32(11008, 4096)], m) == "strip nw=16 cpl=1 spw=22 form=dma-A row_tiles=1" + sm
    assert plan(lib, [g32(2112, 4096)], 16) == "strip nw=16 cpl=1 spw=6 form=dma-A row_tiles=1" + sm    # 66 k-steps over 16 waves: 5 -> 6
    assert plan(lib, [g32(2112, 4096)], 4) == "strip nw=16 cpl=1 spw=5 form=register-A row_tiles=1" + sm
    assert plan(lib, [g32(4096, 4096)], 32) == "strip nw=8 cpl=1 spw=16 form=dma-A row_tiles=2" + sm    # (up to 4096 x 4096 and 32 rows: strips; measured on g32 too)
    assert plan(lib, [g32(4096, 4096)], 33) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4" + sm
    assert plan(lib, [g32(4096, 1024)] * 2, 32) == "panel cols=64 row_tiles=2 k_halves=2 split_k=4 layers=2" + sm
    assert plan(lib, [g32(4096, 1024)] * 2, 16) == "strip nw=16 cpl=1 spw=8 form=dma-A row_tiles=1" + sm
    assert plan(lib, [g32(4096, 4096)], 64) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4" + sm
    assert plan(lib, [g32(4096, 4096)], 65) == "gemm2 tile=256x128 split_k=8" + sm   # (eight row tiles of 32-wide groups are not built)
    assert plan(lib, [g32(4096, 1024)] * 2, 64) == "panel cols=64 row_tiles=4 k_halves=2 split_k=4 layers=2" + sm
    assert plan(lib, [g32(4096, 1024)] * 2, 65).startswith("unsupported")
    assert plan(lib, [W(4096, 4096, 32, 3, NATIVE)], 1).startswith("unsupported")       # 3 bits: 64 / 128 only
    assert plan(lib, [W(4096, 4096, 32)], 1).startswith("skinny")                       # reference layouts in place: the split-K kernel
    assert plan(lib, [W(4096, 4096, 256, layout=NATIVE)], 1).startswith("unsupported")  # group sizes the strips do not serve
    assert plan(lib, [W(128, 4096, layout=NATIVE)], 1).startswith("unsupported")        # K shorter than one round of 8 k-steps


def test_native_layout_validation_and_sizes(lib):
    I, U = _lib.QLLM_ERR_INVALID, _lib.QLLM_ERR_UNSUPPORTED
    sz = [C.c_size_t(0) for _ in range(3)]
    refs = [C.byref(z) for z in sz]
    assert lib.qllm_native_sizes(C.byref(W(4096, 11008, layout=AWQ)), *refs) == 0, _lib.last_error()
    assert [z.value for z in sz] == [4096 * 11008 // 2, 32 * 11008 * 2, 32 * (11008 // 16) * 8]
    assert lib.qllm_native_sizes(C.byref(W(4096, 4096, 64, 3, HQQ)), *refs) == 0
    assert [z.value for z in sz] == [4096 * 3 // 32 * 4096 * 4, 64 * 4096 * 2, 64 * 4096 * 2]
    assert lib.qllm_native_sizes(C.byref(W(4096, 4096, zeros=None)), *refs) == 0 and sz[2].value == 0   # symmetric: no zero points
    for bad in (W(4096, 4096, bits=8), W(4096, 4104), W(4100, 4096), W(4096, 4096, 48), W(4096, 4112, 128, 3)):
        assert lib.qllm_native_sizes(C.byref(bad), *refs) == U, (bad.K, bad.N, bad.bits, bad.group_size)
    assert lib.qllm_native_sizes(None, *refs) == I
    # repack / unpack argument checks (nothing is launched: every call fails validation first)
    assert lib.qllm_repack_native(C.byref(W(4096, 4096, g_idx=16)), 32, 32, 32, None) == U and "act-order" in _lib.last_error()
    assert lib.qllm_repack_native(C.byref(W(4096, 4096, layout=NATIVE)), 32, 32, 32, None) == I
    assert lib.qllm_repack_native(C.byref(W(4096, 4096)), None, 32, 32, None) == I
    assert lib.qllm_repack_native(C.byref(W(4096, 4096)), 36, 32, 32, None) == I                     # alignment
    assert lib.qllm_unpack_native(C.byref(W(4096, 4096)), GPTQ, 32, 32, 32, None) == I               # source must be native
    assert lib.qllm_unpack_native(C.byref(W(4096, 4096, layout=NATIVE)), HQQ, 32, 32, 32, None) == I  # packed zeros <-> GPTQ / AWQ
    assert lib.qllm_unpack_native(C.byref(W(4096, 4096, layout=NATIVE_F16Z)), GPTQ, 32, 32, 32, None) == I
    assert lib.qllm_unpack_native(C.byref(W(4096, 4096, layout=NATIVE)), 7, 32, 32, 32, None) == I
    # a native descriptor with g_idx is malformed; mixing layout families in one grouped launch is refused
    arr = (_lib.QllmWeight * 2)(W(4096, 4096, layout=NATIVE), W(4096, 4096))
    ys = (C.c_void_p * 2)(64, 128)
    assert lib.qllm_linear_forward_grouped(arr, ys, 2, 256, 1, _lib.DT_F16, None, 0, None) == I and "layout family" in _lib.last_error()
    arr1 = (_lib.QllmWeight * 1)(W(4096, 4096, layout=NATIVE, g_idx=16))
    buf = C.create_string_buffer(256)
    assert lib.qllm_plan_describe(arr1, 1, 1, 1, buf, 256) == I
    assert lib.qllm_dequant(C.byref(W(4096, 4096, layout=NATIVE)), 64, _lib.DT_F16, 0, None) == U


def test_debug_timeline_argument(lib):
    assert lib.qllm_debug_timeline(None, 0) == 0


def test_gather_columns_argument_checks(lib):
    I, U = _lib.QLLM_ERR_INVALID, _lib.QLLM_ERR_UNSUPPORTED
    g = lib.qllm_gather_columns
    assert g(None, 64, 128, 4, 4096, _lib.DT_F16, None) == I
    assert g(64, 64, 64, 4, 4096, _lib.DT_F16, None) == I and "alias" in _lib.last_error()
    assert g(64, 64, 136, 4, 4096, _lib.DT_F16, None) == I                           # 16-byte alignment
    assert g(64, 128, 256, 4, 4096, 7, None) == I                                    # element type
    assert g(64, 128, 256, -1, 4096, _lib.DT_F16, None) == I
    assert g(64, 128, 256, 4, 4100, _lib.DT_F16, None) == U                          # K % 8
    assert g(64, 128, 256, 4, 32768, _lib.DT_BF16, None) == U                        # K beyond the LDS row buffers
    assert g(64, 128, 256, 0, 4096, _lib.DT_F16, None) == 0                          # empty batch: nothing launched


def test_release_library_has_no_reachable_one_row_tile_panel(lib, monkeypatch):
    """Round 5: below 17 rows every native layer is on the strips -- the shipped library is the release build (its knobs are compile-time
    constants: an environment variable cannot bring the round-4 route back) and announces no panel plan there, whatever the shape."""
    assert lib.qllm_is_lab_build() == 0
    monkeypatch.setenv("QLLM_PANEL_MIN_M", "9")
    for K, N, g, bits, lay in ((11008, 4096, 128, 4, NATIVE), (11008, 4096, 64, 4, NATIVE_F16Z), (11008, 4096, 64, 3, NATIVE_F16Z),
                               (28672, 8192, 128, 4, NATIVE), (4096, 4096, 32, 4, NATIVE), (8192, 1024, 128, 4, NATIVE)):
        for m in range(2, 17):
            assert plan(lib, [W(K, N, g, bits, lay)], m).startswith(("strip ", "strip1 ")), (K, N, g, bits, m)   # (strip1: batches 2..4, round 6)
        assert plan(lib, [W(K, N, g, bits, lay)], 33).startswith("panel "), (K, N, g, bits)


def test_planner_thresholds_can_be_moved_at_run_time(lib):
    """Round-5 verdict, weak #9: the decision tree's thresholds were compile-time constants in the release build.  qllm_set_knob
    (ABI 6) moves the settable ones; qllm_plan_describe -- which asks the decision functions the forward calls execute -- follows."""
    from qllm_amd import ops
    sm = " layout=strip-major"
    attn, up = W(4096, 4096, layout=NATIVE), W(4096, 11008, layout=NATIVE)
    try:
        assert ops.get_knob("QLLM_STRIP1") is None
        assert plan(lib, [attn], 1).startswith("strip1 ")
        ops.set_knob("QLLM_STRIP1", 0)
        assert ops.get_knob("QLLM_STRIP1") == 0
        assert plan(lib, [attn], 1) == "strip nw=8 cpl=1 spw=16 form=lds-slab row_tiles=1" + sm      # the round-4 batch-1 path
        assert plan(lib, [up] * 2, 17).startswith("panel ")
        ops.set_knob("QLLM_PANEL_GROUP_MIN_M", 33)
        assert plan(lib, [up] * 2, 17).startswith("strip ") and plan(lib, [up] * 2, 33).startswith("panel ")
        assert plan(lib, [W(4096, 4096)], 1024).startswith("gemm3 ")
        ops.set_knob("QLLM_GEMM3_MIN_M", 2048)
        assert plan(lib, [W(4096, 4096)], 1024).startswith("gemm2 ") and plan(lib, [W(4096, 4096)], 2048).startswith("gemm3 ")
        ops.set_knob("QLLM_GEMM3", 0)
        assert plan(lib, [W(4096, 4096)], 2048).startswith("gemm2 ")
        # names and values outside what every built kernel covers are refused, and change nothing
        with pytest.raises(Exception, match="not a settable"):
            ops.set_knob("QLLM_GEMM4", 1)
        with pytest.raises(Exception, match="17..129"):
            ops.set_knob("QLLM_PANEL_MIN_M", 9)
    finally:
        ops.reset_knobs()
    assert ops.get_knob("QLLM_STRIP1") is None and plan(lib, [attn], 1).startswith("strip1 ")
    assert plan(lib, [W(4096, 4096)], 1024).startswith("gemm3 ")


def test_ragged_last_round_of_tiles_is_split_over_k(lib):
    """Round 6 (profiles/r06_shape_table.md): more 256x128 tiles than CUs with a last round that fills at most half of them -- the tiles of
    that round are shared by 2 / 4 / 8 blocks each (gemm3.hip, tail split); Llama-2-7B's own shapes are unchanged."""
    g3 = "gemm3 tile=256x128 matrix-waves=8 staging-waves=4"
    sm = " layout=strip-major"
    assert plan(lib, [W(5120, 5120)], 2048) == g3 + " tail_split=4"              # Llama-2-13B: 320 tiles = 256 + 64 x 4
    assert plan(lib, [W(5120, 5120, layout=NATIVE)], 2048) == g3 + " tail_split=4" + sm
    assert plan(lib, [W(13824, 5120)], 2048) == g3 + " tail_split=4"
    assert plan(lib, [W(5120, 13824)], 2048) == g3 + " tail_split=2"             # 864 tiles = 768 + 96 x 2
    assert plan(lib, [W(4096, 14336)], 2048) == g3 + " tail_split=2"             # Llama-3-8B / Mistral-7B: 896 = 768 + 128 x 2
    assert plan(lib, [W(4096, 4096)], 2304) == g3 + " tail_split=4"              # 288 tiles: 32 x 8 would leave 8 k-tiles per block
    assert plan(lib, [W(5120, 5120)], 2048, have_ws=0) == g3                     # no workspace: no split, still fused
    assert plan(lib, [W(4096, 11008)], 2048) == g3 and plan(lib, [W(11008, 4096)], 2048) == g3   # 688 = 512 + 176: nothing to split
    assert plan(lib, [W(8192, 28672)], 2048) == g3                               # 1792 tiles: seven whole rounds
    from qllm_amd import ops
    try:
        ops.set_knob("QLLM_GEMM3_TAIL", 0)
        assert plan(lib, [W(5120, 5120)], 2048) == g3
    finally:
        ops.reset_knobs()
    # the workspace a caller is told to bring covers the tail's partial tiles (64 tiles x 4 blocks x 128 KB) + the counters
    arr = (_lib.QllmWeight * 1)(W(5120, 5120))
    assert lib.qllm_workspace_bytes_act(arr, 2048, _lib.DT_F16) >= 16384 + 64 * 4 * 256 * 128 * 4


def test_prefill_sized_groups_share_one_launch_of_the_256x128_kernel(lib):
    """Round 6: q/k/v and gate/up at prefill sizes as ONE grid of the wave-specialised kernel (gemm3.hip, grouped form): the rounds of CUs
    are counted over the group -- Llama-2-7B's gate/up are 1376 tiles (5.4 rounds, the last one K-split) instead of 2 x 688 (2 x 3)."""
    from qllm_amd import ops
    g3 = "gemm3 tile=256x128 matrix-waves=8 staging-waves=4"
    sm = " layout=strip-major"
    attn, up = W(4096, 4096, layout=NATIVE), W(4096, 11008, layout=NATIVE)
    assert plan(lib, [up] * 2, 2048) == g3 + " layers=2 tail_split=2" + sm          # 1376 = 5 x 256 + 96 x 2
    assert plan(lib, [attn] * 3, 2048) == g3 + " layers=3" + sm                      # 768 tiles: three whole rounds
    assert plan(lib, [attn, W(4096, 1024, layout=NATIVE), W(4096, 1024, layout=NATIVE)], 2048) == g3 + " layers=3 tail_split=2" + sm  # GQA: 384 tiles
    assert plan(lib, [W(4096, 4096)] * 3, 2048) == g3 + " layers=3"                   # the reference's row-stream buffers in place too
    assert plan(lib, [up] * 2, 2048, have_ws=0) == g3 + " layers=2" + sm             # no workspace: no K split, still one launch
    assert plan(lib, [up] * 2, 384) == g3 + " layers=2 tail_split=2" + sm            # 344 tiles = 256 + 88 x 2
    for m in (129, 383):
        assert plan(lib, [up] * 2, m).startswith("unsupported")                       # below 384 rows: layer by layer (gemm2 / panel)
    assert plan(lib, [W(4096, 1024, layout=NATIVE)] * 2, 512).startswith("unsupported")   # 32 tiles: fewer than CUs -> single launches split K
    assert plan(lib, [W(4096, 4096, layout=AWQ)] * 3, 2048).startswith("unsupported")     # AWQ words in place: no grouped form
    assert plan(lib, [W(4096, 4096, 128, 3, NATIVE)] * 3, 2048).startswith("unsupported")
    try:
        ops.set_knob("QLLM_GEMM3_GROUP", 0)
        assert plan(lib, [up] * 2, 2048).startswith("unsupported")
    finally:
        ops.reset_knobs()
