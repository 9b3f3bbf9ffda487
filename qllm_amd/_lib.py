"""ctypes binding of libqllm_mi355x.so (the C ABI declared in include/qllm_mi355x.h).

This is the only place the shared library is touched.  There is no CPU / eager fallback: if the library is missing
or the device is not gfx950, every hot-path entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libqllm_mi355x.so"
LIB_PATH = os.environ.get("QLLM_MI355X_LIB") or os.path.join(_HERE, LIB_NAME)  # override: A/B-testing kernel builds

QLLM_OK, QLLM_ERR_INVALID, QLLM_ERR_UNSUPPORTED, QLLM_ERR_WORKSPACE, QLLM_ERR_LAUNCH, QLLM_ERR_DEVICE = range(6)
LAYOUT_GPTQ, LAYOUT_AWQ_GEMM, LAYOUT_HQQ, LAYOUT_NATIVE, LAYOUT_NATIVE_F16Z = 0, 1, 2, 3, 4
DT_F16, DT_BF16, DT_F16_IN_BF16_OUT = 0, 1, 2
ABI_VERSION = 6

EXPORTS = (
    "qllm_abi_version", "qllm_is_lab_build", "qllm_last_error", "qllm_device_info", "qllm_workspace_bytes", "qllm_workspace_bytes_act", "qllm_workspace_init",
    "qllm_linear_forward", "qllm_linear_forward_grouped", "qllm_dequant", "qllm_ort_gemv", "qllm_ort_dequant",
    "qllm_awq_gemm_forward", "qllm_unpack_qweight", "qllm_pack_qweight", "qllm_gather_columns", "qllm_ort_dequantize4bits",
    "qllm_plan_describe", "qllm_debug_timeline", "qllm_native_sizes", "qllm_repack_native", "qllm_unpack_native",
    "qllm_comm_buffer_bytes", "qllm_comm_alloc", "qllm_comm_free", "qllm_comm_export", "qllm_comm_import", "qllm_comm_close",
    "qllm_allreduce_oneshot", "qllm_linear_forward_allreduce", "qllm_convert_bf16_to_f16",
    "qllm_set_knob", "qllm_get_knob", "qllm_reset_knobs",
)


class QllmWeight(C.Structure):
    """struct qllm_weight (include/qllm_mi355x.h)."""
    _fields_ = [
        ("qweight", C.c_void_p), ("scales", C.c_void_p), ("qzeros", C.c_void_p), ("g_idx", C.c_void_p),
        ("bias", C.c_void_p),
        ("K", C.c_int32), ("N", C.c_int32), ("group_size", C.c_int32), ("bits", C.c_int32),
        ("layout", C.c_int32), ("add_zero_bias", C.c_int32),
    ]


class QllmDeviceInfo(C.Structure):
    _fields_ = [
        ("arch", C.c_char * 32), ("compute_units", C.c_int32), ("wavefront_size", C.c_int32),
        ("lds_bytes_per_cu", C.c_int32), ("clock_khz", C.c_int32), ("hbm_bytes", C.c_int64),
    ]


class QllmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"qllm_mi355x error {code}: {msg}")
        self.code = code


class QllmUnsupported(QllmError):
    pass


_lib = None
_lock = threading.Lock()


def _declare(lib):
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    wp = C.POINTER(QllmWeight)
    lib.qllm_abi_version.restype = C.c_int
    lib.qllm_abi_version.argtypes = []
    lib.qllm_is_lab_build.restype = C.c_int
    lib.qllm_is_lab_build.argtypes = []
    lib.qllm_set_knob.restype = C.c_int
    lib.qllm_set_knob.argtypes = [C.c_char_p, i32]
    lib.qllm_get_knob.restype = C.c_int
    lib.qllm_get_knob.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32)]
    lib.qllm_reset_knobs.restype = None
    lib.qllm_reset_knobs.argtypes = []
    lib.qllm_last_error.restype = C.c_char_p
    lib.qllm_last_error.argtypes = []
    lib.qllm_device_info.restype = C.c_int
    lib.qllm_device_info.argtypes = [C.c_int, C.POINTER(QllmDeviceInfo)]
    lib.qllm_workspace_bytes.restype = sz
    lib.qllm_workspace_bytes.argtypes = [wp, i32]
    lib.qllm_workspace_bytes_act.restype = sz
    lib.qllm_workspace_bytes_act.argtypes = [wp, i32, i32]
    lib.qllm_comm_buffer_bytes.restype = sz
    lib.qllm_comm_buffer_bytes.argtypes = [i32, sz]
    lib.qllm_comm_alloc.restype = C.c_int
    lib.qllm_comm_alloc.argtypes = [sz, C.POINTER(vp)]
    lib.qllm_comm_free.restype = C.c_int
    lib.qllm_comm_free.argtypes = [vp]
    lib.qllm_comm_export.restype = C.c_int
    lib.qllm_comm_export.argtypes = [vp, vp]
    lib.qllm_comm_import.restype = C.c_int
    lib.qllm_comm_import.argtypes = [vp, C.POINTER(vp)]
    lib.qllm_comm_close.restype = C.c_int
    lib.qllm_comm_close.argtypes = [vp]
    lib.qllm_allreduce_oneshot.restype = C.c_int
    lib.qllm_allreduce_oneshot.argtypes = [vp, i32, i32, vp, i32, i32, sz, vp, vp]
    lib.qllm_linear_forward_allreduce.restype = C.c_int
    lib.qllm_linear_forward_allreduce.argtypes = [C.POINTER(QllmWeight), vp, vp, i32, i32, vp, i32, i32, sz, vp, vp]
    lib.qllm_convert_bf16_to_f16.restype = C.c_int
    lib.qllm_convert_bf16_to_f16.argtypes = [vp, vp, sz, vp]
    lib.qllm_workspace_init.restype = C.c_int
    lib.qllm_workspace_init.argtypes = [vp, sz, vp]
    lib.qllm_linear_forward.restype = C.c_int
    lib.qllm_linear_forward.argtypes = [wp, vp, vp, i32, i32, vp, sz, vp]
    lib.qllm_linear_forward_grouped.restype = C.c_int
    lib.qllm_linear_forward_grouped.argtypes = [wp, C.POINTER(vp), i32, vp, i32, i32, vp, sz, vp]
    lib.qllm_dequant.restype = C.c_int
    lib.qllm_dequant.argtypes = [wp, vp, i32, i32, vp]
    lib.qllm_ort_gemv.restype = C.c_int
    lib.qllm_ort_gemv.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, vp, sz, vp]
    lib.qllm_ort_dequant.restype = C.c_int
    lib.qllm_ort_dequant.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]
    lib.qllm_awq_gemm_forward.restype = C.c_int
    lib.qllm_awq_gemm_forward.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, sz, vp]
    lib.qllm_unpack_qweight.restype = C.c_int
    lib.qllm_unpack_qweight.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.qllm_pack_qweight.restype = C.c_int
    lib.qllm_pack_qweight.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.qllm_gather_columns.restype = C.c_int
    lib.qllm_gather_columns.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.qllm_plan_describe.restype = C.c_int
    lib.qllm_plan_describe.argtypes = [wp, i32, i32, i32, C.c_char_p, sz]
    lib.qllm_debug_timeline.restype = C.c_int
    lib.qllm_debug_timeline.argtypes = [vp, i32]
    lib.qllm_native_sizes.restype = C.c_int
    lib.qllm_native_sizes.argtypes = [wp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    lib.qllm_repack_native.restype = C.c_int
    lib.qllm_repack_native.argtypes = [wp, vp, vp, vp, vp]
    lib.qllm_unpack_native.restype = C.c_int
    lib.qllm_unpack_native.argtypes = [wp, i32, vp, vp, vp, vp]
    lib.qllm_ort_dequantize4bits.restype = C.c_int
    lib.qllm_ort_dequantize4bits.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, vp, vp]


def is_built() -> bool:
    return os.path.exists(LIB_PATH)


def load():
    """dlopen the in-tree library (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_NAME} is not built (expected {LIB_PATH}). Run `python -c 'import __graft_entry__ as g; "
                    f"g.build()'` or `make -C qllm_amd/csrc`. qllm_amd has no CPU fallback for the forward path.")
            lib = C.CDLL(LIB_PATH)
            _declare(lib)
            v = lib.qllm_abi_version()
            if v != ABI_VERSION:
                raise RuntimeError(f"{LIB_NAME} ABI version {v} != expected {ABI_VERSION}; rebuild it")
            _lib = lib
    return _lib


def last_error() -> str:
    return (load().qllm_last_error() or b"").decode("utf-8", "replace")


def check(rc: int):
    if rc == QLLM_OK:
        return
    msg = last_error()
    if rc == QLLM_ERR_UNSUPPORTED:
        raise QllmUnsupported(rc, msg)
    raise QllmError(rc, msg)


def device_info(device: int = 0) -> dict:
    info = QllmDeviceInfo()
    check(load().qllm_device_info(int(device), C.byref(info)))
    return dict(arch=info.arch.decode(), compute_units=info.compute_units, wavefront_size=info.wavefront_size,
                lds_bytes_per_cu=info.lds_bytes_per_cu, clock_khz=info.clock_khz, hbm_bytes=info.hbm_bytes)
