"""Self-contained registries with the type strings the reference configs use.

The reference registers its classes into the OpenMMLab registries (mmdet ``HEADS``/``TRANSFORMER``/``BBOX_CODERS``,
mmcv ``ATTENTION``/``TRANSFORMER_LAYER``/``TRANSFORMER_LAYER_SEQUENCE``/``POSITIONAL_ENCODING``).  mmcv/mmdet are not
installed on the GPU box, so the build carries registries of its own that accept the ``roi_head=dict(...)`` config
subtree verbatim; when a real mmcv/mmdet IS importable, ``mirror_into_openmmlab()`` additionally registers the same
classes there (``force=True``) so that ``plugin_dir``-style flows (tools/test.py:155-165) pick up this implementation.
"""
import copy


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self.module_dict[key] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'{self.name}: cfg must be a dict with a "type" key, got {cfg!r}')
        args = copy.deepcopy(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        t = args.pop('type')
        cls = t if isinstance(t, type) else self.module_dict.get(t)
        if cls is None:
            raise KeyError(f'{t} is not in the {self.name} registry')
        return cls(**args)

    def __contains__(self, key):
        return key in self.module_dict


HEADS = Registry('head')
TRANSFORMER = Registry('Transformer')
TRANSFORMER_LAYER = Registry('transformerLayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
ATTENTION = Registry('attention')
FEEDFORWARD_NETWORK = Registry('feed-forward Network')
POSITIONAL_ENCODING = Registry('position encoding')
BBOX_CODERS = Registry('bbox_coder')
ROI_EXTRACTORS = Registry('roi_extractor')
LOSSES = Registry('loss')
NECKS = Registry('neck')
BBOX_ASSIGNERS = Registry('bbox_assigner')


def build_from_cfg(cfg, registry, default_args=None):
    return registry.build(cfg, default_args)


def build_head(cfg, train_cfg=None, test_cfg=None):
    """mmdet ``build_head`` as used by the detector shell (mmdet3d_plugin/models/detectors/mv2d.py:34-38)."""
    default = {}
    if train_cfg is not None:
        default['train_cfg'] = train_cfg
    if test_cfg is not None:
        default['test_cfg'] = test_cfg
    return HEADS.build(cfg, default or None)


build_roi_extractor = ROI_EXTRACTORS.build
build_bbox_coder = BBOX_CODERS.build
build_transformer = TRANSFORMER.build
build_transformer_layer = TRANSFORMER_LAYER.build
build_transformer_layer_sequence = TRANSFORMER_LAYER_SEQUENCE.build
build_attention = ATTENTION.build
build_positional_encoding = POSITIONAL_ENCODING.build
build_loss = LOSSES.build
build_neck = NECKS.build


def mirror_into_openmmlab():
    """Best effort: register this implementation's classes into real mmcv / mmdet registries when present."""
    done = []
    try:
        from mmdet.models.builder import HEADS as MM_HEADS, NECKS as MM_NECKS   # type: ignore
        from mmdet.models.utils.builder import TRANSFORMER as MM_TR           # type: ignore
        from mmdet.core.bbox.builder import BBOX_CODERS as MM_CODERS          # type: ignore
        from mmcv.cnn.bricks.registry import (ATTENTION as MM_ATT, TRANSFORMER_LAYER as MM_TL,            # type: ignore
                                              TRANSFORMER_LAYER_SEQUENCE as MM_TLS, POSITIONAL_ENCODING as MM_PE)
    except Exception:
        return done
    # NECKS is deliberately NOT mirrored: mmdet's own FPN must keep serving the 2-D detector (only the single-level MV2D neck is built here)
    _ = MM_NECKS
    for src, dst in ((HEADS, MM_HEADS), (TRANSFORMER, MM_TR), (BBOX_CODERS, MM_CODERS), (ATTENTION, MM_ATT),
                     (TRANSFORMER_LAYER, MM_TL), (TRANSFORMER_LAYER_SEQUENCE, MM_TLS), (POSITIONAL_ENCODING, MM_PE)):
        for name, cls in src.module_dict.items():
            dst.register_module(name=name, force=True, module=cls)
            done.append(name)
    return done
