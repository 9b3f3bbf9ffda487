"""Checkpoint loader / saver for quantized HF causal-LM directories -- SURVEY.md section 8(f) row 1.

Mirrors what the reference's `AutoQuantizedModelForCausalLM.from_quantized / save_pretrained`
(qllm/modeling/base.py:226-336) and `BaseQuantizeConfig` (qllm/modeling/config.py:81-125) do for the layouts this build
serves, for LOCAL directories (no hub access here): parse the quantisation config, build the empty model from its HF
config, swap the quantized nn.Linear modules for q_layers, load the safetensors shards into the unchanged buffer
names, apply the AutoGPTQ `qzeros+1` normalisation.  Everything on the per-token path stays in the q_layers.
"""
from __future__ import annotations

import glob
import json
import os
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from ..utils import modelutils


@dataclass
class QuantConfig:
    """quant_config.json / quantize_config.json / config.json["quantization_config"] (config.py:81-119)."""
    bits: int = 4
    group_size: int = 128
    version: str = "GPTQ"            # pack mode: GPTQ | GEMM | HQQ | ORT
    quant_method: str = "gptq"
    compatible_with_autogptq: bool = False
    by_layer: Dict[str, dict] = field(default_factory=dict)  # quant_config_by_layer.json (mixed precision)

    @classmethod
    def from_dir(cls, path: str) -> "QuantConfig":
        raw = None
        for name in ("quant_config.json", "quantize_config.json"):
            f = os.path.join(path, name)
            if os.path.exists(f):
                raw = json.load(open(f))
                break
        if raw is None:
            f = os.path.join(path, "config.json")
            if os.path.exists(f):
                raw = json.load(open(f)).get("quantization_config")
                if raw is not None and raw.get("use_exllama", False):
                    raise ValueError("use_exllama is not supported yet")
        if raw is None:
            raise FileNotFoundError("quant_config.json/quantize_config.json not found in checkpoint directory")
        bits = raw.get("w_bit", raw.get("bits"))
        group = raw.get("q_group_size", raw.get("group_size"))
        if bits is None or group is None:
            raise ValueError("quantisation config needs bits/w_bit and group_size/q_group_size")
        cfg = cls(bits=int(bits), group_size=int(group))
        cfg.compatible_with_autogptq = bool(raw.get("COMPATIBLE_WITH_AUTOGPTQ", False))
        if "version" not in raw:  # GPTQ-for-LLaMa / AutoGPTQ checkpoints: GPTQ layout, zeros stored minus one
            cfg.version, cfg.quant_method, cfg.compatible_with_autogptq = "GPTQ", "gptq", True
        else:
            cfg.version = str(raw["version"]).upper()
            cfg.quant_method = raw.get("quant_method", "awq")
        f = os.path.join(path, "quant_config_by_layer.json")
        if os.path.exists(f):
            cfg.by_layer = {k: v for k, v in json.load(open(f)).items() if isinstance(v, dict)}
        return cfg

    def to_dict(self) -> dict:
        d = dict(bits=self.bits, group_size=self.group_size, version=self.version, quant_method=self.quant_method)
        if self.compatible_with_autogptq:
            d["COMPATIBLE_WITH_AUTOGPTQ"] = 1
        return d


def _checkpoint_files(path: str):
    idx = glob.glob(os.path.join(path, "*.safetensors.index.json")) + glob.glob(os.path.join(path, "*.bin.index.json"))
    if idx:
        wm = json.load(open(idx[0]))
        wm = wm.get("weight_map", wm)
        return sorted({os.path.join(path, v) for v in wm.values()})
    files = sorted(glob.glob(os.path.join(path, "*.safetensors"))) or sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not files:
        raise ValueError(f"{path} is not a folder containing weights or safetensors")
    return files


def _load_file(f: str):
    if f.endswith(".safetensors"):
        import safetensors.torch
        return safetensors.torch.load_file(f, device="cpu")
    return torch.load(f, map_location="cpu", weights_only=True)


def _no_init_weights():
    """transformers moved no_init_weights between releases (modeling_utils -> initialization); weights are about to be
    overwritten by the checkpoint, so skipping the random init is only a speed-up -- fall back to a no-op."""
    import contextlib
    import transformers
    for mod in ("initialization", "modeling_utils"):
        m = getattr(transformers, mod, None)
        if m is None:
            try:
                m = __import__(f"transformers.{mod}", fromlist=["x"])
            except Exception:  # noqa: BLE001
                continue
        if hasattr(m, "no_init_weights"):
            return m.no_init_weights()
    return contextlib.nullcontext()


def swap_quantized_linears(model: torch.nn.Module, quantized_names, cfg: QuantConfig):
    """nn.Linear -> q_layer for every name in `quantized_names` (base.py:281-286 + modelutils.py:161-181)."""
    target = modelutils.select_quant_linear(cfg.version, cfg.bits, cfg.quant_method)
    info = {n: cfg.by_layer.get(n, {"wbits": cfg.bits, "groupsize": cfg.group_size}) for n in quantized_names}
    modelutils.make_mixbits_quant_linear(model, set(quantized_names), info, target_layer=target)
    # q/k/v and gate/up read the same tensor: one grouped launch per group at decode sizes (q_layers/fused.py); the modules,
    # their names and their state-dict buffers are untouched
    from .q_layers import install_sibling_groups
    model.sibling_groups = install_sibling_groups(model, [target])
    return target


def release_reference_layouts(model: torch.nn.Module, on: bool = True) -> int:
    """Memory policy of the q_layers (q_layers/_hip_forward.py): once a layer's native copy exists on the device, drop the packed
    reference buffers it duplicates (regenerated bit-exactly for state_dict / save_pretrained / unpack / .to()), so that a loaded
    model costs 1.0x its checkpoint's bytes of HBM instead of 2.0x.  Returns the number of layers the policy was set on."""
    from .q_layers import QuantLinearGPTQ, QuantLinearHQQ, QuantLinearORT, WQLinear_GEMM
    n = 0
    for m in model.modules():
        if isinstance(m, (QuantLinearGPTQ, QuantLinearHQQ, QuantLinearORT, WQLinear_GEMM)):
            m.release_reference = bool(on)
            n += 1
    return n


def load_quantized(model_dir: str, device: Optional[str] = "cuda", torch_dtype: Optional[torch.dtype] = None,
                   release_reference: bool = True):
    """Local equivalent of AutoQuantizedModelForCausalLM.from_quantized (base.py:226-322).  `release_reference`: see
    release_reference_layouts (QLLM_RELEASE_REFERENCE=0 in the environment keeps both copies)."""
    import transformers

    hf_cfg = transformers.AutoConfig.from_pretrained(model_dir)
    dtype = torch_dtype or getattr(hf_cfg, "torch_dtype", None) or torch.float16
    cfg = QuantConfig.from_dir(model_dir)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with _no_init_weights():
            model = transformers.AutoModelForCausalLM.from_config(hf_cfg)
    finally:
        torch.set_default_dtype(prev)
    files = _checkpoint_files(model_dir)
    # un-quantized layers (lm_head ...) are detected by the absence of a `.qweight` key (base.py:272-275)
    keys = set()
    shards = []
    for f in files:
        sd = _load_file(f)
        keys.update(sd.keys())
        shards.append(sd)
    linears = modelutils.find_layers(model, [torch.nn.Linear])
    quantized = [n for n in linears if n + ".qweight" in keys]
    target = swap_quantized_linears(model, quantized, cfg)
    model.tie_weights()
    missing, unexpected = [], []
    for sd in shards:
        # g_idx is a plain attribute on AWQ/HQQ layers: drop it there, like strict=False does for the reference
        ret = model.load_state_dict(sd, strict=False)
        missing.extend(ret.missing_keys)
        unexpected.extend(k for k in ret.unexpected_keys if not k.endswith(".bias") and not k.endswith(".g_idx"))
    model.quant_config = cfg
    if cfg.compatible_with_autogptq and cfg.version == "GPTQ":
        for _, layer in modelutils.find_layers(model, [target]).items():
            layer.handle_qzeros_for_autogptq()
        cfg.compatible_with_autogptq = False  # zeros are now stored plainly (base.py:314-319)
    model.load_report = dict(quantized_layers=len(quantized), unexpected_keys=unexpected)
    if device is not None:
        model = model.to(device)
    if release_reference and os.environ.get("QLLM_RELEASE_REFERENCE", "1") != "0":
        release_reference_layouts(model)
    return model.eval()


def save_quantized(model: torch.nn.Module, save_dir: str, cfg: Optional[QuantConfig] = None):
    """safetensors + quantize_config.json + quant_config_by_layer.json (base.py:325-336)."""
    cfg = cfg or getattr(model, "quant_config", None) or QuantConfig()
    os.makedirs(save_dir, exist_ok=True)
    model.config.quantization_config = cfg.to_dict()
    model.save_pretrained(save_dir, safe_serialization=True)
    from ..modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM
    by_layer = {n: {"wbits": l.bits, "groupsize": l.groupsize}
                for n, l in modelutils.find_layers(model, [QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM]).items()}
    json.dump(by_layer, open(os.path.join(save_dir, "quant_config_by_layer.json"), "w"), indent=4)
    json.dump(cfg.to_dict(), open(os.path.join(save_dir, "quantize_config.json"), "w"), indent=4)
