"""Drop-in boundary on CPU: the reference ``roi_head=dict(...)`` config subtrees (CFG-S:40-121, CFG-T:40-125, restated in
mv2d_amd/configs.py) build through the build's registries, every type string of SURVEY.md §8(b) resolves, and the
modules expose exactly the reference ``state_dict`` key layout (so released checkpoints load unchanged)."""
import numpy as np
import pytest
import torch

import mv2d_amd
from mv2d_amd import configs, registry, synthetic


@pytest.mark.parametrize('cfg_fn,cls_name', [(configs.roi_head_cfg_s, 'MV2DSHead'), (configs.roi_head_cfg_t, 'MV2DTHead')])
def test_reference_config_builds_and_state_dict_layout(cfg_fn, cls_name):
    head = mv2d_amd.build_head(cfg_fn(), test_cfg=configs.TEST_CFG_RCNN)
    assert type(head).__name__ == cls_name
    ref_keys = set(synthetic.make_head_state(seed=0).keys())
    keys = set(head.state_dict().keys())
    assert keys == ref_keys, (sorted(keys - ref_keys)[:5], sorted(ref_keys - keys)[:5])
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}
    missing, unexpected = head.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    n_params = sum(v.numel() for v in head.state_dict().values())
    assert n_params == 11253378 + 1518339 + 1248256          # head + query generator + PE


def test_every_reference_type_string_resolves():
    R = registry
    for reg, names in [(R.HEADS, ['MV2DHead', 'MV2DSHead', 'MV2DTHead', 'CrossAttentionBoxHead', 'QueryGenerator']),
                       (R.TRANSFORMER, ['MV2DTransformer', 'PETRTransformer', 'PETRDNTransformer']),
                       (R.TRANSFORMER_LAYER, ['PETRTransformerDecoderLayer']),
                       (R.TRANSFORMER_LAYER_SEQUENCE, ['PETRTransformerDecoder', 'PETRTransformerEncoder']),
                       (R.ATTENTION, ['FlattenMHSelfAttention', 'PETRMultiheadAttention']),
                       (R.POSITIONAL_ENCODING, ['SinePositionalEncoding3D', 'LearnedPositionalEncoding3D']),
                       (R.BBOX_CODERS, ['NMSFreeCoder']), (R.ROI_EXTRACTORS, ['SingleRoIExtractor']),
                       (R.LOSSES, ['FocalLoss', 'L1Loss'])]:
        for n in names:
            assert n in reg, f'{n} missing from {reg.name}'


def test_head_signatures_match_reference():
    import inspect
    from mv2d_amd.plugin.heads import CrossAttentionBoxHead, MV2DHead, MV2DSHead, MV2DTHead
    from mv2d_amd.plugin.modules import FlattenMHSelfAttention, PETRMultiheadAttention
    p = inspect.signature(MV2DHead.__init__).parameters
    assert list(p)[1:12] == ['bbox_roi_extractor', 'bbox_head', 'query_generator', 'pe', 'box_correlation', 'pc_range',
                             'intrins_feat_scale', 'feat_lvl', 'force_fp32', 'train_cfg', 'test_cfg']
    assert p['intrins_feat_scale'].default == 0.1 and p['feat_lvl'].default == 0 and p['force_fp32'].default is False
    assert list(inspect.signature(MV2DHead.simple_test).parameters) == ['self', 'x', 'proposal_list', 'img_metas', 'rescale']
    ps = inspect.signature(MV2DSHead.__init__).parameters
    assert [ps[k].default for k in ('use_denoise', 'neg_bbox_loss', 'denoise_scalar', 'denoise_noise_scale', 'denoise_noise_trans',
                                   'denoise_weight', 'denoise_split')] == [False, False, 10, 1.0, 0.0, 1.0, 0.75]
    assert inspect.signature(MV2DTHead.__init__).parameters['num_views'].default == 6
    pf = inspect.signature(CrossAttentionBoxHead.forward).parameters
    assert list(pf)[1:10] == ['reference_points', 'x', 'masks', 'pos_embed', 'attn_mask', 'cross_attn_mask', 'force_fp32',
                              'query_embeds', 'return_query_feats']
    for cls in (FlattenMHSelfAttention, PETRMultiheadAttention):
        pa = inspect.signature(cls.forward).parameters
        assert list(pa)[1:9] == ['query', 'key', 'value', 'identity', 'query_pos', 'key_pos', 'attn_mask', 'key_padding_mask']


def test_product_path_has_no_cpu_fallback():
    """simple_test on CPU tensors must fail loudly (the engine asserts GPU input) instead of running something else."""
    head = mv2d_amd.build_head(configs.roi_head_cfg_t(), test_cfg=configs.TEST_CFG_RCNN).eval()
    prob = synthetic.make_problem('micro_t', seed=0)
    with pytest.raises(Exception):
        head.simple_test([torch.from_numpy(prob['feat'])], [torch.from_numpy(p) for p in prob['proposals']], prob['img_metas'])


def test_product_does_not_import_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.abspath(mv2d_amd.__file__))
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f'{f} imports the oracle'


def test_neck_registry_and_scope():
    """f2: the single-level FPN of the MV2D configs builds through the registry with mmdet's state-dict keys; anything else fails loudly."""
    import mv2d_amd
    neck = mv2d_amd.build_neck(dict(type='FPN', in_channels=[256] * 5, out_channels=256, start_level=2, end_level=2, num_outs=1))
    assert set(neck.state_dict()) == {'lateral_convs.0.conv.weight', 'lateral_convs.0.conv.bias', 'fpn_convs.0.conv.weight', 'fpn_convs.0.conv.bias'}
    assert neck.state_dict()['fpn_convs.0.conv.weight'].shape == (256, 256, 3, 3)
    import pytest
    with pytest.raises(NotImplementedError):
        mv2d_amd.build_neck(dict(type='FPN', in_channels=[256] * 5, out_channels=256, num_outs=5))
