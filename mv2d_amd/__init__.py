"""mv2d_amd — MI355X (gfx950) native implementation of MV2D's sparse cross-attention decoder hot path.

Importing the package registers the reference's type strings (MV2DHead / MV2DSHead / MV2DTHead,
CrossAttentionBoxHead, QueryGenerator, MV2DTransformer, PETRTransformer*, FlattenMHSelfAttention,
PETRMultiheadAttention, SinePositionalEncoding3D, NMSFreeCoder, ...) in mv2d_amd.registry, and — when a real
mmcv/mmdet is importable — mirrors them into the OpenMMLab registries so reference configs resolve to this code.
"""
from . import registry  # noqa: F401
from .plugin import heads, modules, neck  # noqa: F401
from .registry import build_head, build_neck  # noqa: F401

__all__ = ['registry', 'build_head', 'build_neck']

registry.mirror_into_openmmlab()
