"""decompdiff_amd — MI355X-native reverse-diffusion sampling hot path of bytedance/DecompDiff.

Public surface mirrors what the reference's sampling script touches
(/root/reference/scripts/sample_diffusion_decomp.py:25,537-544,329-360):

    from decompdiff_amd import DecompScorePosNet3D, log_sample_categorical

The heavy lifting is in libdecompdiff_hip.so (decompdiff_amd/csrc, C ABI in
include/decompdiff_hip.h); build it with ``python -m decompdiff_amd.build``.
"""
from .config import ModelConfig, shipped_config  # noqa: F401
from .model import DecompScorePosNet3D, log_sample_categorical  # noqa: F401

__all__ = ["DecompScorePosNet3D", "log_sample_categorical", "ModelConfig", "shipped_config"]
