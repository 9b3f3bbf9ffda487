"""torch-facing wrappers over the C ABI: pass data_ptr()s + the current HIP stream, allocate outputs and the
split-K workspace with torch's caching allocator (the library itself never allocates).

PyTorch is plumbing here (device memory, streams); all compute is in libqllm_mi355x.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (CHAIN_POLL_X, CHAIN_PUBLISH_Y, DT_BF16, DT_F16, LAYOUT_AWQ_GEMM, LAYOUT_GPTQ, LAYOUT_HQQ, QllmError,
                   QllmUnsupported, QllmWeight)

LAYOUTS = {"GPTQ": LAYOUT_GPTQ, "GEMM": LAYOUT_AWQ_GEMM, "AWQ": LAYOUT_AWQ_GEMM, "HQQ": LAYOUT_HQQ}

_workspaces: dict = {}


def _check_input(t: torch.Tensor, name: str):
    # the reference's CHECK_INPUT (csrc/ort_cuda/ort_ops.cc:6-10): device tensor, contiguous
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP/CUDA tensor (qllm_amd has no CPU forward path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _act_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return DT_F16
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise RuntimeError(f"activations must be float16 or bfloat16, got {t.dtype}")


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Per-(device, stream) scratch for split-K slabs + arrival counters; zero-filled once (kernels leave it clean).
    The key uses the CURRENT stream of `device` (not of whatever device happens to be current)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        # 64 MB up front covers every configuration of the Llama-class shapes (largest: 32 MB of split-K partial tiles):
        # growing later would move the buffer under hipGraphs captured with the old pointer
        size = max(int(nbytes), 64 << 20)
        ws = torch.zeros(size, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def make_weight(layout: str, qweight, scales, qzeros, g_idx, bias, in_features: int, out_features: int,
                group_size: int, bits: int, add_zero_bias: int = 0):
    """Build the C descriptor.  Returns (QllmWeight, keepalive tuple of tensors whose pointers it holds)."""
    lay = LAYOUTS[layout.upper()]
    for name, t in (("qweight", qweight), ("scales", scales)):
        _check_input(t, name)
    if scales.dtype != torch.float16:
        raise RuntimeError("scales must be float16 at the C boundary (cast bf16 scales once, like ort_ops.cc:79-90)")
    if qweight.dtype != torch.int32:
        raise RuntimeError("qweight must be int32")
    if qzeros is not None:
        _check_input(qzeros, "qzeros")
        want = torch.float16 if lay == LAYOUT_HQQ else torch.int32
        if qzeros.dtype != want:
            raise RuntimeError(f"qzeros must be {want} for layout {layout}")
    if g_idx is not None:
        _check_input(g_idx, "g_idx")
        if g_idx.dtype != torch.int32:
            raise RuntimeError("g_idx must be int32")
    if bias is not None:
        _check_input(bias, "bias")
        if bias.dtype != torch.float16:
            raise RuntimeError("bias must be float16 at the C boundary")
    w = QllmWeight(
        qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr() if qzeros is not None else None,
        g_idx.data_ptr() if g_idx is not None else None, bias.data_ptr() if bias is not None else None,
        int(in_features), int(out_features), int(group_size), int(bits), lay, int(add_zero_bias))
    return w, (qweight, scales, qzeros, g_idx, bias)


# ---- chained decode -------------------------------------------------------------------------
_active_chain: dict = {}


class DecodeChain:
    """Back-to-back decode-sized quantized linears (M <= 4) issued as a CHAIN of links that alternate between two side
    streams, so that link i+1 is resident and has all of its weight loads in flight while link i still computes; the
    activation vector travels in-band (include/qllm_mi355x.h, qllm_linear_forward_chained).

        chain = ops.DecodeChain(device)          # once; owns two streams, an activation arena and an error word
        with chain:                              # per decode step; capturable in a hipGraph
            y = model_stack(x)                   # q_layer forwards / ops.linear_forward[_grouped] inside take the chain

    Inside the context every eligible forward allocates its outputs from the chain's arena (pre-filled with 0xFF bytes at
    `__enter__`), so outputs returned to the caller stay valid only until the chain is entered again -- the contract of a
    graph's static outputs.  A forward the chain cannot take (shape without a chained plan, M > 4) joins both streams and
    runs as an ordinary launch.  torch ops on chained outputs must come after the `with` block (or after `chain.join()`)."""

    def __init__(self, device=None, arena_bytes: int = 8 << 20, mode: Optional[str] = None):
        # "engine": the links of a step are RECORDED and run as one persistent launch (csrc/engine.hip: a loader wave per CU
        #           streams the weights through an LDS ring, consumer waves wait only for activations);
        # "streams": every link is its own launch, alternating between two streams (csrc/strip.hip, CH variants).
        self.mode = mode or os.environ.get("QLLM_CHAIN_MODE", "streams")  # measured: streams 865 tok/s, engine 775 (profiles/r02_engine.md)
        if self.mode not in ("engine", "streams"):
            raise ValueError("DecodeChain mode must be 'engine' or 'streams'")
        self._prog: list = []          # engine mode: recorded links of the current segment
        self._keep: list = []          # ... and the tensors they name, alive until the segment is launched (a recorded link holds
                                       #     raw pointers; an input freed before the flush could be handed out again)
        self._prog_cache: dict = {}    # program bytes -> device copy
        self._strip0 = 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.device):
            self.streams = (torch.cuda.Stream(), torch.cuda.Stream())
            self.arena = torch.full((arena_bytes,), 0xFF, dtype=torch.uint8, device=self.device)  # armed once, entirely
            self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._lo = self.arena.data_ptr()
        self._hi = self._lo + arena_bytes
        self._cursor = 0
        self._high = 0       # high-water mark of the arena over all steps so far: only [0, _high) can hold stale results
        self._armed = 0      # extent re-armed (filled with 0xFF) at the current __enter__
        self._turn = 0
        self._main = None
        self._live = self.streams
        self.links = 0       # launches taken as chained links since the last __enter__
        self.fallbacks = 0   # launches that had to join and run as ordinary launches

    # -- context ------------------------------------------------------------------------------------------------------
    def __enter__(self):
        idx = self.device.index
        if _active_chain.get(idx) is not None:
            raise RuntimeError("a DecodeChain is already active on this device")
        self._main = torch.cuda.current_stream(idx)
        # re-arm what earlier steps wrote: every half of every chained output = 0xFFFF ("not written yet").  On the caller's
        # stream, ahead of the fork; bytes beyond the high-water mark were never written since construction.
        self._armed = self._high
        self._capturing = torch.cuda.is_current_stream_capturing()
        if self._capturing and self._armed == 0:
            self._armed = self.arena.numel()  # captured without a warm-up step: the replayed fill must cover whatever is used
        if self._armed:
            with torch.cuda.device(self.device):
                self.arena[:self._armed].fill_(0xFF)
        self._cursor = 0
        self._turn = 0
        self.links = 0
        self.fallbacks = 0
        self._prog = []
        self._keep = []
        self._strip0 = 0
        if self.mode == "engine":
            self._live = (self._main, self._main)
        elif os.environ.get("QLLM_CHAIN_SERIAL", "0") == "1":
            # profiling aid: the same chained kernels, all on the caller's stream (no overlap; every poll succeeds at once) --
            # counter-collecting profilers serialise dispatches, under which an overlapped chain would sit out its time-outs
            self._live = (self._main, self._main)
        else:
            self._live = self.streams
        for s in self._live:
            if s is not self._main:
                s.wait_stream(self._main)
        _active_chain[idx] = self
        return self

    def __exit__(self, *exc):
        _active_chain.pop(self.device.index, None)
        self._high = max(self._high, self._cursor)
        self.join()
        if self._capturing and self._cursor > self._armed and (not exc or exc[0] is None):
            raise RuntimeError("DecodeChain: the captured step used more of the arena than the warm-up steps before it, so its "
                               "replays would not re-arm those bytes; run the same step once eagerly before capturing")
        self._main = None
        return False

    def join(self):
        """Order the caller's stream after everything issued (or recorded) in the chain so far."""
        if self.mode == "engine":
            self._flush()
            return
        for s in self._live:
            if s is not self._main:
                self._main.wait_stream(s)

    def _flush(self):
        """engine mode: launch the links recorded since the last flush as one persistent program on the caller's stream."""
        if not self._prog:
            return
        n = len(self._prog)
        arr = (_lib.QllmEngineLink * n)(*self._prog)
        key = bytes(arr)
        dev = self._prog_cache.get(key)
        if dev is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("DecodeChain(engine): this step's program is not on the device yet and cannot be uploaded inside "
                                   "a graph capture; run the same step once eagerly before capturing")
            dev = torch.frombuffer(bytearray(key), dtype=torch.uint8).to(self.device)
            self._prog_cache[key] = dev
        with torch.cuda.device(self.device):
            rc = _lib.load().qllm_engine_run(dev.data_ptr(), n, self.err.data_ptr(), self._main.cuda_stream)
        _lib.check(rc)
        self._prog = []
        self._keep = []  # (stream-ordered reuse after the launch is safe: it runs on the caller's stream)

    # -- used by the forward wrappers -----------------------------------------------------------------------------------
    def owns(self, t: torch.Tensor) -> bool:
        return self._lo <= t.data_ptr() < self._hi

    def alloc(self, shape, dtype) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        start = (self._cursor + 255) // 256 * 256
        if start + nbytes > self.arena.numel():
            raise RuntimeError(f"DecodeChain arena exhausted ({self.arena.numel()} bytes): construct it with a larger arena_bytes")
        self._cursor = start + nbytes
        return self.arena[start:start + nbytes].view(dtype).view(*shape)

    def next_stream(self) -> torch.cuda.Stream:
        s = self._live[self._turn]
        self._turn ^= 1
        return s

    def check(self):
        """Host-synchronising: raise if any link's poll loop gave up (a producer never wrote its outputs)."""
        if int(self.err.item()) != 0:
            self.err.zero_()
            raise QllmError(_lib.QLLM_ERR_LAUNCH, "a chained decode link timed out waiting for its input (DecodeChain.err != 0)")


def active_chain(device: torch.device) -> Optional[DecodeChain]:
    return _active_chain.get(device.index if device.index is not None else torch.cuda.current_device())


def _engine_record(chain: DecodeChain, ws_desc: Sequence[QllmWeight], x2d: torch.Tensor):
    """engine mode: validate + record the layers (one link each; layers that share x just name the same input).  None when
    one of them is outside the engine's scope: nothing is recorded then."""
    lib = _lib.load()
    if x2d.shape[0] != 1:
        return None
    cursor, strip0, n_prog = chain._cursor, chain._strip0, len(chain._prog)
    poll = 1 if chain.owns(x2d) else 0
    outs = []
    for w in ws_desc:
        y = chain.alloc((1, w.N), x2d.dtype)
        link = _lib.QllmEngineLink()
        rc = lib.qllm_engine_link_init(C.byref(w), x2d.data_ptr(), y.data_ptr(), 1, _act_dtype(x2d), poll, chain._strip0, C.byref(link))
        if rc == _lib.QLLM_ERR_UNSUPPORTED:
            chain._cursor, chain._strip0 = cursor, strip0
            del chain._prog[n_prog:]
            return None
        _lib.check(rc)
        chain._prog.append(link)
        chain._strip0 += w.N // 32
        outs.append(y)
    chain._keep.append(x2d)
    chain.links += 1
    return outs


def _chained_forward(chain: DecodeChain, ws_desc: Sequence[QllmWeight], x2d: torch.Tensor):
    """One chained link, or None when the library has no chained plan for it (the caller then joins and launches normally)."""
    if chain.mode == "engine":
        return _engine_record(chain, ws_desc, x2d)
    lib = _lib.load()
    n, m = len(ws_desc), x2d.shape[0]
    arr = (QllmWeight * n)(*ws_desc)
    buf = C.create_string_buffer(128)
    _lib.check(lib.qllm_chain_plan_describe(arr, n, m, buf, 128))
    if not buf.value.startswith(b"chained"):
        return None
    cursor = chain._cursor
    outs = [chain.alloc((m, w.N), x2d.dtype) for w in ws_desc]
    ys = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    flags = CHAIN_PUBLISH_Y
    stream = chain.next_stream()
    # (arena bytes beyond what __enter__ re-armed were never written since construction, when all of it was armed)
    if chain.owns(x2d):
        flags |= CHAIN_POLL_X
    elif stream is not chain._main:
        stream.wait_stream(chain._main)  # x came from ordinary work on the caller's stream
    with torch.cuda.device(x2d.device):
        rc = lib.qllm_linear_forward_chained(arr, ys, n, x2d.data_ptr(), m, _act_dtype(x2d), flags, chain.err.data_ptr(),
                                             stream.cuda_stream)
    if rc == _lib.QLLM_ERR_UNSUPPORTED:
        chain._cursor = cursor
        chain._turn ^= 1
        return None
    _lib.check(rc)
    chain.links += 1
    return outs


def _check_x(x2d: torch.Tensor, ws_desc: Sequence[QllmWeight]):
    _check_input(x2d, "x")
    if x2d.dim() != 2:
        raise RuntimeError(f"x must be 2-D [M, K], got {tuple(x2d.shape)}")
    for w in ws_desc:
        if x2d.shape[1] != w.K:
            raise RuntimeError(f"x must be [M, {w.K}], got {tuple(x2d.shape)}")


def _check_out(o: torch.Tensor, m: int, n_cols: int, x2d: torch.Tensor):
    if o.device != x2d.device or o.dtype != x2d.dtype or tuple(o.shape) != (m, n_cols) or not o.is_contiguous():
        raise RuntimeError(f"out must be a contiguous [{m}, {n_cols}] {x2d.dtype} tensor on {x2d.device}, "
                           f"got {tuple(o.shape)} {o.dtype} on {o.device}")


def linear_forward(w: QllmWeight, x2d: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[M,N] = x2d[M,K] . dequant(w) (+bias) through the fused HIP kernels.  Raises QllmUnsupported when the
    library has no fused kernel for the configuration (caller may then use dequant() + matmul)."""
    _check_x(x2d, (w,))
    lib = _lib.load()
    m = x2d.shape[0]
    if out is not None:
        _check_out(out, m, w.N, x2d)
    if m == 0:
        return out if out is not None else torch.empty((0, w.N), dtype=x2d.dtype, device=x2d.device)
    chain = active_chain(x2d.device)
    if chain is not None and out is None:
        outs = _chained_forward(chain, (w,), x2d) if m <= 4 else None
        if outs is not None:
            return outs[0]
        chain.join()
        chain.fallbacks += 1
    if out is None:
        out = torch.empty((m, w.N), dtype=x2d.dtype, device=x2d.device)
    with torch.cuda.device(x2d.device):
        nbytes = lib.qllm_workspace_bytes(C.byref(w), m)
        ws = workspace(x2d.device, nbytes)
        rc = lib.qllm_linear_forward(C.byref(w), x2d.data_ptr(), out.data_ptr(), m, _act_dtype(x2d), ws.data_ptr(),
                                     ws.numel(), _stream_ptr())
    _lib.check(rc)
    return out


def linear_forward_grouped(ws_desc: Sequence[QllmWeight], x2d: torch.Tensor,
                           outs: Optional[Sequence[torch.Tensor]] = None):
    """Several layers sharing x (q/k/v, gate/up) in ONE launch (decode sizes only)."""
    _check_x(x2d, ws_desc)
    lib = _lib.load()
    n = len(ws_desc)
    m = x2d.shape[0]
    if outs is not None:
        if len(outs) != n:
            raise RuntimeError(f"outs must hold {n} tensors, got {len(outs)}")
        for o, w in zip(outs, ws_desc):
            _check_out(o, m, w.N, x2d)
    if m == 0:
        return list(outs) if outs is not None else [torch.empty((0, w.N), dtype=x2d.dtype, device=x2d.device) for w in ws_desc]
    chain = active_chain(x2d.device)
    if chain is not None and outs is None:
        got = _chained_forward(chain, ws_desc, x2d) if m <= 4 else None
        if got is not None:
            return got
        chain.join()
        chain.fallbacks += 1
    if outs is None:
        outs = [torch.empty((m, w.N), dtype=x2d.dtype, device=x2d.device) for w in ws_desc]
    arr = (QllmWeight * n)(*ws_desc)
    ys = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    with torch.cuda.device(x2d.device):
        nbytes = sum(lib.qllm_workspace_bytes(C.byref(w), m) for w in ws_desc)
        wsp = workspace(x2d.device, nbytes)
        rc = lib.qllm_linear_forward_grouped(arr, ys, n, x2d.data_ptr(), m, _act_dtype(x2d), wsp.data_ptr(),
                                             wsp.numel(), _stream_ptr())
    _lib.check(rc)
    return list(outs)


def dequant(w: QllmWeight, device: torch.device, dtype=torch.float16, transposed: bool = False) -> torch.Tensor:
    """W[K,N] (or [N,K]) bit-identical to the reference's DequantizeLinearBlockWise / unpack()."""
    lib = _lib.load()
    shape = (w.N, w.K) if transposed else (w.K, w.N)
    out = torch.empty(shape, dtype=dtype, device=device)
    dt = DT_F16 if dtype == torch.float16 else DT_BF16
    if dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("dequant output must be float16 or bfloat16")
    with torch.cuda.device(device):
        rc = lib.qllm_dequant(C.byref(w), out.data_ptr(), dt, 1 if transposed else 0, _stream_ptr())
    _lib.check(rc)
    return out


def chain_plan_describe(ws_desc: Sequence[QllmWeight], m: int) -> str:
    """The chained-link plan of these descriptors at m rows ("chained strip ..."), or "not chainable"."""
    arr = (QllmWeight * len(ws_desc))(*ws_desc)
    buf = C.create_string_buffer(256)
    _lib.check(_lib.load().qllm_chain_plan_describe(arr, len(ws_desc), int(m), buf, 256))
    return buf.value.decode()


def plan_describe(ws_desc: Sequence[QllmWeight], m: int, have_workspace: bool = True) -> str:
    """Which kernel `linear_forward` (one descriptor) / `linear_forward_grouped` (several) would run for m rows."""
    arr = (QllmWeight * len(ws_desc))(*ws_desc)
    buf = C.create_string_buffer(256)
    _lib.check(_lib.load().qllm_plan_describe(arr, len(ws_desc), int(m), 1 if have_workspace else 0, buf, 256))
    return buf.value.decode()


def ort_dequantize4bits(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, g_idx: Optional[torch.Tensor],
                        block_size: int, in_features: int, out_features: int) -> torch.Tensor:
    """ORT / MatMulNBits blob -> W[N,K] in the scales' dtype (ort_ops.Dequantize4Bits, ort_ops.cc:161-197)."""
    for name, t in (("qweight", qweight), ("scales", scales), ("qzeros", qzeros)):
        _check_input(t, name)
    if qweight.dtype != torch.uint8:
        raise RuntimeError("qweight must be uint8 (ORT blob layout)")
    zf = qzeros.dtype != torch.uint8
    s16 = scales if scales.dtype == torch.float16 else scales.to(torch.float16)
    z = qzeros if (not zf or qzeros.dtype == torch.float16) else qzeros.to(torch.float16)
    gi = None
    if g_idx is not None:
        gi = g_idx.to(device=qweight.device, dtype=torch.int32).contiguous()
    out = torch.empty((out_features, in_features), dtype=torch.float16, device=qweight.device)
    with torch.cuda.device(qweight.device):
        rc = _lib.load().qllm_ort_dequantize4bits(qweight.data_ptr(), s16.data_ptr(), z.data_ptr(), 1 if zf else 0,
                                                  gi.data_ptr() if gi is not None else None, block_size, in_features,
                                                  out_features, out.data_ptr(), _stream_ptr())
    _lib.check(rc)
    return out if scales.dtype == torch.float16 else out.to(scales.dtype)


def gather_columns(x2d: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """x2d[:, perm] for a contiguous [M, K] fp16 / bf16 matrix and an int32 device permutation (act-order layers: the
    activation side of the row-sorted weight copy).  The library's LDS-staged gather; shapes it does not take
    (K % 8, K > 28672) fall back to index_select."""
    _check_x(x2d, ())
    if perm.dtype != torch.int32 or perm.device != x2d.device or perm.numel() != x2d.shape[1] or not perm.is_contiguous():
        raise RuntimeError("perm must be a contiguous int32 tensor of K entries on x's device")
    out = torch.empty_like(x2d)
    if x2d.shape[0] == 0:
        return out
    with torch.cuda.device(x2d.device):
        rc = _lib.load().qllm_gather_columns(x2d.data_ptr(), perm.data_ptr(), out.data_ptr(), x2d.shape[0], x2d.shape[1],
                                             _act_dtype(x2d), _stream_ptr())
    if rc == _lib.QLLM_ERR_UNSUPPORTED:
        return x2d.index_select(1, perm.long())
    _lib.check(rc)
    return out


def unpack_qweight(qweight: torch.Tensor, layout: str, bits: int, in_features: int, out_features: int) -> torch.Tensor:
    _check_input(qweight, "qweight")
    q = torch.empty((in_features, out_features), dtype=torch.int32, device=qweight.device)
    with torch.cuda.device(qweight.device):
        rc = _lib.load().qllm_unpack_qweight(qweight.data_ptr(), LAYOUTS[layout.upper()], bits, in_features,
                                             out_features, q.data_ptr(), _stream_ptr())
    _lib.check(rc)
    return q


def pack_qweight(q_kn: torch.Tensor, layout: str, bits: int) -> torch.Tensor:
    _check_input(q_kn, "q_kn")
    if q_kn.dtype != torch.int32:
        raise RuntimeError("q_kn must be int32")
    k, n = q_kn.shape
    lay = LAYOUTS[layout.upper()]
    shape = (k, n // 8) if lay == LAYOUT_AWQ_GEMM else (k * bits // 32, n)
    out = torch.empty(shape, dtype=torch.int32, device=q_kn.device)
    with torch.cuda.device(q_kn.device):
        rc = _lib.load().qllm_pack_qweight(q_kn.data_ptr(), lay, bits, k, n, out.data_ptr(), _stream_ptr())
    _lib.check(rc)
    return out


__all__ = ["make_weight", "linear_forward", "linear_forward_grouped", "dequant", "gather_columns", "unpack_qweight", "pack_qweight",
           "workspace", "QllmUnsupported", "LAYOUTS", "DecodeChain", "active_chain", "plan_describe", "chain_plan_describe"]
