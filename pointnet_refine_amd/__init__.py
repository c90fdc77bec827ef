"""MI355X-native LineRefineNet hot path (shared-MLP context encoder, pooling,
context_proj, line point-MLP) behind the reference's nn.Module surface.

    from pointnet_refine_amd.model import LineRefineNet   # drop-in for src.model
"""
from .model import (DetrTransformerDecoderLayer, LineRefineNet, MultiScalePointNetEncoder,  # noqa: F401
                    PositionalEncoding)

__all__ = ["LineRefineNet", "MultiScalePointNetEncoder", "PositionalEncoding",
           "DetrTransformerDecoderLayer"]
