"""Dispatch helpers on the boundary (reference: qllm/utils/modelutils.py:17-24, 44-68, 128-140, 161-181):
choose the q_layer class for a (pack_mode, wbits, quant_method) triple and swap nn.Linear modules for it."""
from __future__ import annotations

import torch.nn as nn


def find_layers(module, layers=(nn.Conv2d, nn.Linear), name=""):
    if layers is None or type(module) in tuple(layers):
        return {name: module}
    res = {}
    for child_name, child in module.named_children():
        res.update(find_layers(child, layers=layers, name=name + "." + child_name if name != "" else child_name))
    return res


def select_quant_linear(pack_mode: str, wbits: int, quant_method: str):
    """Same decision table as the reference (modelutils.py:44-68) restricted to the layouts this build serves:
    ORT -> QuantLinearORT; hqq -> QuantLinearHQQ; GEMM, or AUTO on a 4-bit-capable engine -> WQLinear_GEMM; otherwise
    QuantLinearGPTQ.  MARLIN / vptq are outside the hot-path scope and raise."""
    from ..modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, QuantLinearORT, WQLinear_GEMM
    from ..modeling.q_layers.ext_package_checker import is_the_machine_support_awq_engine

    pack_mode = pack_mode.upper()
    quant_method = quant_method.lower()
    if quant_method == "vptq":
        raise NotImplementedError("quant_method=vptq is outside this build's scope")
    if pack_mode == "ORT":
        return QuantLinearORT
    if quant_method == "hqq":
        return QuantLinearHQQ
    if pack_mode == "MARLIN":
        raise NotImplementedError("pack_mode=MARLIN is outside this build's scope")
    if pack_mode == "GEMM" or (pack_mode == "AUTO" and is_the_machine_support_awq_engine(wbits)):
        return WQLinear_GEMM
    return QuantLinearGPTQ


def set_op_by_name(layer, name, new_module):
    levels = name.split(".")
    mod = layer
    for lvl in levels[:-1]:
        mod = mod[int(lvl)] if lvl.isdigit() else getattr(mod, lvl)
    setattr(mod, levels[-1], new_module)


def get_op_by_name(module, op_name):
    for name, m in module.named_modules():
        if name == op_name:
            return m
    raise ValueError(f"Cannot find op {op_name} in module {module}")


def make_mixbits_quant_linear(module, replaced_names, quant_info: dict, name="", target_layer=None):
    """Swap every nn.Linear named in `replaced_names` for `target_layer(bits, groupsize, in, out, bias, dtype=)`;
    per-layer (wbits, groupsize) come from quant_info[name] unless quant_info carries global ones
    (reference modelutils.py:161-181)."""
    dtype = next(iter(module.parameters())).dtype
    for module_name, sub_module in list(module.named_modules()):
        if module_name not in replaced_names:
            continue
        if "groupsize" in quant_info and "wbits" in quant_info:
            bits, groupsize = quant_info["wbits"], quant_info["groupsize"]
        else:
            bits, groupsize = quant_info[module_name]["wbits"], quant_info[module_name]["groupsize"]
        new_module = target_layer(bits, groupsize, sub_module.in_features, sub_module.out_features,
                                  sub_module.bias is not None, dtype=dtype)
        new_module.bias = sub_module.bias.data if sub_module.bias is not None else None
        set_op_by_name(module, module_name, new_module)
