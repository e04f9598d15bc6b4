"""lemevit_amd -- MI355X-native (gfx950) LeMeViT backbone hot path.

Importing the package loads the HIP kernel library through ctypes and raises if it is missing:
there is no CPU or eager-PyTorch fallback for the hot path.

    from lemevit_amd import create_model
    model = create_model("lemevit_base", num_classes=1000).cuda()
"""
from . import _lib  # noqa: F401  (fails loudly when liblemevit_hip.so is absent)
from . import ops  # noqa: F401
from .model import CrossAttention, DualCrossAttention, DualCrossAttention_v2, LeMeBlock, LeMeViT, LeMeViTBackbone, StandardAttention  # noqa: F401
from .optim import FlatAdamW, ModelEma  # noqa: F401
from . import dist  # noqa: F401  (wrap_ddp, FlatGradSync, attach_flat_grad_sync)
from .registry import create_model, is_model, list_models, load_checkpoint, register_model  # noqa: F401
from .registry import lemevit_base, lemevit_small, lemevit_small_v2, lemevit_tiny, lemevit_tiny_v2, vit_tiny  # noqa: F401

__all__ = ["create_model", "register_model", "list_models", "is_model", "load_checkpoint", "LeMeViT", "LeMeViTBackbone", "LeMeBlock",
           "StandardAttention", "DualCrossAttention", "DualCrossAttention_v2", "CrossAttention", "lemevit_tiny", "lemevit_small", "lemevit_base",
           "lemevit_small_v2", "lemevit_tiny_v2", "vit_tiny", "ops", "FlatAdamW", "ModelEma"]
