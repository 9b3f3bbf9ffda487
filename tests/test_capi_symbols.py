"""The C-ABI library builds, loads on a GPU-less host and exports every symbol include/qllm_mi355x.h declares.
No compute is launched here."""
import ctypes
import os
import re

import pytest

from qllm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qllm_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qllm_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not _lib.is_built():
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_every_declared_symbol_is_exported(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name
        assert ctypes.cast(getattr(lib, name), ctypes.c_void_p).value


def test_version_and_error_plumbing(lib):
    assert lib.qllm_abi_version() == _lib.ABI_VERSION
    # argument validation runs before any device work, so it is testable without a GPU
    w = _lib.QllmWeight(None, None, None, None, None, 256, 128, 128, 4, 0, 0)
    rc = lib.qllm_linear_forward(ctypes.byref(w), None, None, 1, 0, None, 0, None)
    assert rc == _lib.QLLM_ERR_INVALID and "NULL" in _lib.last_error()
    w = _lib.QllmWeight(16, 16, None, None, None, 256, 128, 128, 9, 0, 0)
    assert lib.qllm_dequant(ctypes.byref(w), 16, 0, 0, None) == _lib.QLLM_ERR_INVALID
    assert "bits" in _lib.last_error()
    w = _lib.QllmWeight(16, 16, 16, None, None, 256, 100, 128, 4, _lib.LAYOUT_AWQ_GEMM, 0)
    assert lib.qllm_dequant(ctypes.byref(w), 16, 0, 0, None) == _lib.QLLM_ERR_INVALID
    assert "pack_num" in _lib.last_error()
    with pytest.raises(_lib.QllmError):
        _lib.check(_lib.QLLM_ERR_INVALID)


def test_workspace_bytes_is_pure(lib):
    w = _lib.QllmWeight(16, 16, 16, None, None, 4096, 4096, 128, 4, 0, 0)
    b1 = lib.qllm_workspace_bytes(ctypes.byref(w), 1)
    b16 = lib.qllm_workspace_bytes(ctypes.byref(w), 16)
    assert b1 >= 16384 and b16 > b1
    # 256 tiles of 256x128: one block per CU, no split-K; + the fp16 copy of x the wave-specialised kernel reads when x is bf16
    assert lib.qllm_workspace_bytes(ctypes.byref(w), 2048) == 16384 + 2048 * 4096 * 2
    # mid-size M: split-K partial tiles.  M=512: 64 tiles x S=4 (the largest S with tiles*S <= 256 CUs, >= 8 k-tiles each)
    tile = 256 * 128 * 4
    assert lib.qllm_workspace_bytes(ctypes.byref(w), 512) == 16384 + 64 * 4 * tile
    assert lib.qllm_workspace_bytes(ctypes.byref(w), 256) == 16384 + 32 * 8 * tile
    assert lib.qllm_workspace_bytes(ctypes.byref(w), 1024) == 16384 + 128 * 2 * tile + 1024 * 4096 * 2
    wide = _lib.QllmWeight(16, 16, 16, None, None, 4096, 11008, 128, 4, 0, 0)
    # 172 tiles: a split would need two rounds of blocks; from M = 384 the wave-specialised kernel serves this shape (bf16 copy of x)
    assert lib.qllm_workspace_bytes(ctypes.byref(wide), 512) == 16384 + 512 * 4096 * 2
    assert lib.qllm_workspace_bytes(ctypes.byref(wide), 320) == 16384
    short = _lib.QllmWeight(16, 16, 16, None, None, 512, 4096, 128, 4, 0, 0)
    assert lib.qllm_workspace_bytes(ctypes.byref(short), 512) == 16384     # 8 k-tiles: too short to split
    ragged = _lib.QllmWeight(16, 16, 16, None, None, 4096, 4000, 128, 4, 0, 0)
    assert lib.qllm_workspace_bytes(ctypes.byref(ragged), 512) == 16384    # N % 128 != 0: the 128x128 kernel, no slabs
    assert 16384 + 64 * 4 * tile <= 64 << 20                               # fits the 64 MB the Python wrapper allocates up front
    down = _lib.QllmWeight(16, 16, 16, None, None, 11008, 4096, 128, 4, 0, 0)
    assert lib.qllm_workspace_bytes(ctypes.byref(down), 2048) <= 64 << 20  # ... also with the bf16 copy of a [2048, 11008] input
    # the dtype-aware form (ABI 4): fp16 callers are not charged the staging copy; bf16 == the conservative form
    F16, BF16 = 0, 1
    assert lib.qllm_workspace_bytes_act(ctypes.byref(w), 2048, F16) == 16384
    assert lib.qllm_workspace_bytes_act(ctypes.byref(w), 2048, BF16) == lib.qllm_workspace_bytes(ctypes.byref(w), 2048)
    assert lib.qllm_workspace_bytes_act(ctypes.byref(w), 1024, F16) == 16384 + 128 * 2 * tile
    # counters + slabs alone (everything but the staging copy) never exceed the wrapper's fixed 64 MB, at any M
    for M in (33, 64, 65, 128, 256, 384, 512, 1024, 4096, 16384):
        for ww in (w, wide, down):
            assert lib.qllm_workspace_bytes_act(ctypes.byref(ww), M, F16) <= 33 << 20


def test_device_probe_fails_cleanly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    info = _lib.QllmDeviceInfo()
    assert lib.qllm_device_info(0, ctypes.byref(info)) == _lib.QLLM_ERR_DEVICE


def test_comm_entry_points_validate_before_any_device_work(lib):
    """One-shot all-reduce (csrc/comm.hip): the staging-buffer size is a pure function, and every argument check of the launch
    entry point runs before the kernel is enqueued."""
    slot = 16384
    ctl = lib.qllm_comm_buffer_bytes(1, 0)            # the control block alone: flags [2][16] + epoch
    assert ctl >= (2 * 16 + 1) * 4
    for world in (1, 2, 8, 16):
        assert lib.qllm_comm_buffer_bytes(world, slot) == 2 * world * slot + ctl
    fake = ctypes.c_void_p(16)                        # never dereferenced: the calls below are refused first
    call = lambda peers, rank, world, x, n, dt, sl: lib.qllm_allreduce_oneshot(peers, rank, world, x, n, dt, sl, None, None)  # noqa: E731
    assert call(None, 0, 2, fake, 8192, _lib.DT_F16, slot) == _lib.QLLM_ERR_INVALID and "NULL" in _lib.last_error()
    assert call(fake, 2, 2, fake, 8192, _lib.DT_F16, slot) == _lib.QLLM_ERR_INVALID and "rank" in _lib.last_error()
    assert call(fake, 0, 17, fake, 8192, _lib.DT_F16, slot) == _lib.QLLM_ERR_INVALID
    assert call(fake, 0, 2, fake, 8192, 7, slot) == _lib.QLLM_ERR_INVALID and "act_dtype" in _lib.last_error()
    assert call(fake, 0, 2, fake, 8190, _lib.DT_F16, slot) == _lib.QLLM_ERR_UNSUPPORTED          # not a multiple of 8 elements
    assert call(fake, 0, 2, fake, 16384, _lib.DT_BF16, slot) == _lib.QLLM_ERR_UNSUPPORTED        # 32 KB does not fit a 16 KB slot
    assert call(fake, 0, 2, ctypes.c_void_p(24), 8192, _lib.DT_F16, slot) == _lib.QLLM_ERR_INVALID and "aligned" in _lib.last_error()
    assert lib.qllm_comm_alloc(0, ctypes.byref(ctypes.c_void_p())) == _lib.QLLM_ERR_INVALID
    assert lib.qllm_comm_export(None, None) == _lib.QLLM_ERR_INVALID
    assert lib.qllm_comm_import(None, None) == _lib.QLLM_ERR_INVALID
    assert lib.qllm_comm_free(None) == _lib.QLLM_OK and lib.qllm_comm_close(None) == _lib.QLLM_OK   # NULL: nothing to do


def test_workspace_bytes_cover_the_panel_kernels_partial_panels(lib):
    """csrc/panel.hip splits K over blocks only if the caller's workspace holds the fp32 partial panels: a caller who sizes the
    workspace with qllm_workspace_bytes() must get the split qllm_plan_describe() announces."""
    import re
    NATIVE = 3
    buf = ctypes.create_string_buffer(256)
    for K, N in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 1024), (1024, 8192)):
        w = _lib.QllmWeight(16, 16, 16, None, None, K, N, 128, 4, NATIVE, 0)
        for M in (9, 16, 17, 32, 33, 64, 65, 128):
            assert lib.qllm_plan_describe(ctypes.byref(w), 1, M, 1, buf, 256) == 0
            plan = buf.value.decode()
            if not plan.startswith("panel "):
                continue
            S = int(re.search(r"split_k=(\d+)", plan).group(1))
            slabs = (N // 64) * S * 4 * (4 if M <= 64 else 8) * 256 * 4 if S > 1 else 0
            assert lib.qllm_workspace_bytes(ctypes.byref(w), M) >= 16384 + slabs, (K, N, M, plan)
