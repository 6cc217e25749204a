"""Stub layer that lets the UNMODIFIED reference hot-path modules import in the build container.

Build-container-only tooling used by ``oracle/gen_golden.py``; it never runs on the GPU box (the
reference does not travel).  mmcv==1.6.1 / mmdet==2.25.1 / mmdet3d==1.0.0 are not installed here, so the
names the reference imports from them are provided as stand-ins registered in ``sys.modules``:

* pure plumbing (registries, BaseModule, identity decorators, builders) — no arithmetic;
* restated third-party arithmetic — "parity unpinned" (SURVEY.md §8(c)): mmcv ``BaseTransformerLayer``
  op-order loop + ``FFN`` + ``MultiheadAttention.__init__`` kwargs (the attention math itself is
  ``torch.nn.MultiheadAttention``, which IS installed), mmcv ``RoIAlign`` (delegates to
  ``oracle.mv2d_oracle.roi_align``), mmdet ``bbox2roi`` / ``inverse_sigmoid``.

``install(reference_root)`` additionally makes ``mmdet3d_plugin.*`` importable file-by-file without
executing the package ``__init__`` chain (which would pull the nuScenes dataset code).
"""
import copy
import math
import sys
import types

import torch
import torch.nn as nn

from . import mv2d_oracle as O


class Registry:
    def __init__(self, name):
        self.name = name
        self.table = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.table[name or cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg, **defaults):
        cfg = dict(cfg)
        for k, v in defaults.items():
            cfg.setdefault(k, v)
        return self.table[cfg.pop('type')](**cfg)


def _identity_decorator(*a, **k):
    return lambda f: f


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


class ModuleList(nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        super().__init__(modules)


class Sequential(nn.Sequential):
    def __init__(self, *a, init_cfg=None):
        super().__init__(*a)


REG = {n: Registry(n) for n in ('ATTENTION', 'TRANSFORMER_LAYER', 'TRANSFORMER_LAYER_SEQUENCE',
                                'POSITIONAL_ENCODING', 'FFN', 'TRANSFORMER', 'HEADS', 'LOSSES', 'BBOX_CODERS')}


def build_dropout(cfg):
    return nn.Dropout(cfg.get('drop_prob', 0.0)) if cfg else nn.Identity()


def build_norm_layer(cfg, n):
    assert cfg['type'] == 'LN'
    return 'ln', nn.LayerNorm(n)


def build_activation_layer(cfg):
    assert cfg['type'] == 'ReLU'
    return nn.ReLU(inplace=cfg.get('inplace', False))


@REG['FFN'].register_module()
class FFN(BaseModule):
    """mmcv.cnn.bricks.transformer.FFN: Linear-ReLU-Dropout-Linear-Dropout + identity."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True),
                 ffn_drop=0.0, dropout_layer=None, add_identity=True, init_cfg=None, **kw):
        super().__init__(init_cfg)
        layers, c = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(Sequential(nn.Linear(c, feedforward_channels), build_activation_layer(act_cfg),
                                     nn.Dropout(ffn_drop)))
            c = feedforward_channels
        layers += [nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = Sequential(*layers)
        self.dropout_layer = build_dropout(dropout_layer)
        self.add_identity = add_identity
        self.embed_dims = embed_dims

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        return (x if identity is None else identity) + self.dropout_layer(out)


@REG['ATTENTION'].register_module()
class MultiheadAttention(BaseModule):
    """mmcv MultiheadAttention constructor only (the reference subclasses override forward)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=dict(type='Dropout', drop_prob=0.0), init_cfg=None, batch_first=False, **kw):
        super().__init__(init_cfg)
        if 'dropout' in kw:
            attn_drop = kw['dropout']
            dropout_layer = dict(dropout_layer)
            dropout_layer['drop_prob'] = kw.pop('dropout')
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kw)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = build_dropout(dropout_layer)


class BaseTransformerLayer(BaseModule):
    """mmcv BaseTransformerLayer: operation_order loop with the post-norm identity rules."""

    def __init__(self, attn_cfgs=None, ffn_cfgs=None, operation_order=None, norm_cfg=dict(type='LN'),
                 init_cfg=None, batch_first=False, **kw):
        super().__init__(init_cfg)
        ffn_cfgs = dict(ffn_cfgs or dict(type='FFN', embed_dims=256, feedforward_channels=1024, num_fcs=2,
                                         ffn_drop=0.0, act_cfg=dict(type='ReLU', inplace=True)))
        for old, new in dict(feedforward_channels='feedforward_channels', ffn_dropout='ffn_drop',
                             ffn_num_fcs='num_fcs').items():
            if old in kw:
                ffn_cfgs[new] = kw[old]
        self.batch_first = batch_first
        num_attn = operation_order.count('self_attn') + operation_order.count('cross_attn')
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        self.num_attn, self.operation_order, self.norm_cfg = num_attn, operation_order, norm_cfg
        self.pre_norm = operation_order[0] == 'norm'
        self.attentions = ModuleList()
        i = 0
        for op in operation_order:
            if op in ('self_attn', 'cross_attn'):
                c = dict(attn_cfgs[i])
                c.setdefault('batch_first', batch_first)
                a = REG['ATTENTION'].build(c)
                a.operation_name = op
                self.attentions.append(a)
                i += 1
        self.embed_dims = self.attentions[0].embed_dims
        ffn_cfgs.setdefault('embed_dims', self.embed_dims)
        self.ffns = ModuleList([REG['FFN'].build(ffn_cfgs) for _ in range(operation_order.count('ffn'))])
        self.norms = ModuleList([build_norm_layer(norm_cfg, self.embed_dims)[1]
                                 for _ in range(operation_order.count('norm'))])

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kw):
        ni = ai = fi = 0
        identity = query
        if attn_masks is None:
            attn_masks = [None] * self.num_attn
        elif isinstance(attn_masks, torch.Tensor):
            attn_masks = [copy.deepcopy(attn_masks) for _ in range(self.num_attn)]
        for op in self.operation_order:
            if op == 'self_attn':
                query = self.attentions[ai](query, query, query, identity if self.pre_norm else None,
                                            query_pos=query_pos, key_pos=query_pos, attn_mask=attn_masks[ai],
                                            key_padding_mask=query_key_padding_mask, **kw)
                ai += 1
                identity = query
            elif op == 'norm':
                query = self.norms[ni](query)
                ni += 1
            elif op == 'cross_attn':
                query = self.attentions[ai](query, key, value, identity if self.pre_norm else None,
                                            query_pos=query_pos, key_pos=key_pos, attn_mask=attn_masks[ai],
                                            key_padding_mask=key_padding_mask, **kw)
                ai += 1
                identity = query
            elif op == 'ffn':
                query = self.ffns[fi](query, identity if self.pre_norm else None)
                fi += 1
        return query


class TransformerLayerSequence(BaseModule):
    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__(init_cfg)
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        self.num_layers = num_layers
        self.layers = ModuleList([REG['TRANSFORMER_LAYER'].build(c) for c in transformerlayers])
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

    def forward(self, query, key, value, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kw):
        for layer in self.layers:
            query = layer(query, key, value, query_pos=query_pos, key_pos=key_pos, attn_masks=attn_masks,
                          query_key_padding_mask=query_key_padding_mask, key_padding_mask=key_padding_mask, **kw)
        return query


def xavier_init(m, gain=1, bias=0, distribution='normal'):
    if getattr(m, 'weight', None) is not None:
        (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(m.weight, gain=gain)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


class ConvModule(nn.Module):
    """mmcv ConvModule with conv_cfg=None, norm_cfg=None: Conv2d(bias=True) + ReLU."""

    def __init__(self, cin, cout, k, padding=0, conv_cfg=None, norm_cfg=None, **kw):
        super().__init__()
        assert norm_cfg is None and conv_cfg is None
        self.conv = nn.Conv2d(cin, cout, k, padding=padding)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.conv(x))


class _LossPlaceholder(nn.Module):
    def __init__(self, use_sigmoid=False, **kw):
        super().__init__()
        self.use_sigmoid = use_sigmoid


for _n in ('FocalLoss', 'L1Loss', 'CrossEntropyLoss', 'SmoothL1Loss'):
    REG['LOSSES'].table[_n] = _LossPlaceholder


class BaseBBoxCoder:
    pass


class SingleRoIExtractor(nn.Module):
    """mmdet SingleRoIExtractor with a single stride == one RoIAlign call on level 0."""

    def __init__(self, roi_layer, out_channels, featmap_strides, **kw):
        super().__init__()
        assert len(featmap_strides) == 1 and roi_layer['type'] == 'RoIAlign'
        self.cfg = (roi_layer['output_size'], 1.0 / featmap_strides[0], roi_layer['sampling_ratio'])
        self.out_channels = out_channels
        self.num_inputs = 1

    def forward(self, feats, rois):
        return O.roi_align(feats[0], rois, *self.cfg)


class BaseRoIHead(BaseModule):
    with_bbox = True

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, train_cfg=None, test_cfg=None, init_cfg=None, **kw):
        super().__init__(init_cfg)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.init_bbox_head(bbox_roi_extractor, bbox_head)
        self.init_assigner_sampler()


def _noop(*a, **k):
    return None


def _mod(name, **attrs):
    m = sys.modules.get(name) or types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    if '.' in name:
        parent, child = name.rsplit('.', 1)
        setattr(_mod(parent), child, m)
    return m


def install(reference_root='/root/reference'):
    sys.dont_write_bytecode = True        # never drop __pycache__ next to the (read-only) reference
    _mod('cv2')
    _mod('mmcv', Config=dict)
    _mod('mmcv.runner', BaseModule=BaseModule, auto_fp16=_identity_decorator, force_fp32=_identity_decorator)
    _mod('mmcv.runner.base_module', BaseModule=BaseModule, ModuleList=ModuleList, Sequential=Sequential)
    _mod('mmcv.utils', ConfigDict=dict, build_from_cfg=lambda cfg, reg, default_args=None: reg.build(cfg, **(default_args or {})),
         deprecated_api_warning=_identity_decorator, to_2tuple=lambda x: (x, x))
    _mod('mmcv.cnn', ConvModule=ConvModule, Conv2d=nn.Conv2d, Linear=nn.Linear,
         build_activation_layer=build_activation_layer, build_conv_layer=_noop, build_norm_layer=build_norm_layer,
         xavier_init=xavier_init, bias_init_with_prob=lambda p: float(-math.log((1 - p) / p)))
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.drop', build_dropout=build_dropout)
    _mod('mmcv.cnn.bricks.registry', ATTENTION=REG['ATTENTION'], TRANSFORMER_LAYER=REG['TRANSFORMER_LAYER'],
         TRANSFORMER_LAYER_SEQUENCE=REG['TRANSFORMER_LAYER_SEQUENCE'])
    _mod('mmcv.cnn.bricks.transformer', BaseTransformerLayer=BaseTransformerLayer,
         TransformerLayerSequence=TransformerLayerSequence, FFN=FFN, MultiheadAttention=MultiheadAttention,
         POSITIONAL_ENCODING=REG['POSITIONAL_ENCODING'],
         build_transformer_layer_sequence=lambda c: REG['TRANSFORMER_LAYER_SEQUENCE'].build(c),
         build_attention=lambda c: REG['ATTENTION'].build(c),
         build_positional_encoding=lambda c: REG['POSITIONAL_ENCODING'].build(c))
    _mod('mmdet')
    _mod('mmdet.core', build_bbox_coder=lambda c: REG['BBOX_CODERS'].build(c), build_assigner=_noop,
         build_sampler=_noop, multi_apply=_noop, reduce_mean=lambda x: x, bbox2roi=O.bbox2roi)
    _mod('mmdet.core.bbox', BaseBBoxCoder=BaseBBoxCoder)
    _mod('mmdet.core.bbox.builder', BBOX_CODERS=REG['BBOX_CODERS'])
    _mod('mmdet.models')
    _mod('mmdet.models.builder', HEADS=REG['HEADS'], build_loss=lambda c: REG['LOSSES'].build(c),
         build_head=lambda c: REG['HEADS'].build(c),
         build_roi_extractor=lambda c: SingleRoIExtractor(**{k: v for k, v in c.items() if k != 'type'}))
    _mod('mmdet.models.losses', accuracy=_noop)
    _mod('mmdet.models.utils', build_transformer=lambda c: REG['TRANSFORMER'].build(c),
         build_linear_layer=lambda cfg, *a, **k: nn.Linear(*a, **k))
    _mod('mmdet.models.utils.builder', TRANSFORMER=REG['TRANSFORMER'])
    _mod('mmdet.models.utils.transformer', inverse_sigmoid=O.inverse_sigmoid)
    _mod('mmdet.models.roi_heads')
    _mod('mmdet.models.roi_heads.base_roi_head', BaseRoIHead=BaseRoIHead)
    _mod('mmdet.models.roi_heads.test_mixins', BBoxTestMixin=type('BBoxTestMixin', (), {}),
         MaskTestMixin=type('MaskTestMixin', (), {}))
    _mod('mmdet3d')
    _mod('mmdet3d.models')
    _mod('mmdet3d.models.builder', HEADS=REG['HEADS'], build_loss=lambda c: REG['LOSSES'].build(c))
    root = reference_root.rstrip('/') + '/mmdet3d_plugin'
    for pkg, sub in [('mmdet3d_plugin', ''), ('mmdet3d_plugin.core', '/core'), ('mmdet3d_plugin.core.bbox', '/core/bbox'),
                     ('mmdet3d_plugin.core.bbox.coders', '/core/bbox/coders'), ('mmdet3d_plugin.models', '/models'),
                     ('mmdet3d_plugin.models.roi_heads', '/models/roi_heads'),
                     ('mmdet3d_plugin.models.roi_heads.bbox_heads', '/models/roi_heads/bbox_heads'),
                     ('mmdet3d_plugin.models.roi_heads.utils', '/models/roi_heads/utils')]:
        _mod(pkg).__path__ = [root + sub]
    import mmdet3d_plugin.models.utils  # noqa: F401  (registers transformer bricks + SinePositionalEncoding3D)
    import mmdet3d_plugin.core.bbox.coders.nms_free_coder  # noqa: F401
    import mmdet3d_plugin.models.roi_heads.bbox_heads.cross_attention_head  # noqa: F401
    from mmdet3d_plugin.models.roi_heads.mv2d_s_head import MV2DSHead
    from mmdet3d_plugin.models.roi_heads.mv2d_t_head import MV2DTHead
    return MV2DSHead, MV2DTHead
