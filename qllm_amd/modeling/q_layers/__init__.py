"""Quantized linear layers with the reference's module contract (qllm/modeling/q_layers/*.py), forward on MI355X."""
from .quant_linear_gptq import QuantLinearGPTQ  # noqa: F401
from .quant_linear_awq import WQLinear_GEMM  # noqa: F401
from .quant_linear_hqq import QuantLinearHQQ  # noqa: F401
from .quant_linear_onnxruntime import QuantLinearORT  # noqa: F401
from .fused import SiblingGroup, fuse_siblings, install_sibling_groups  # noqa: F401
