"""The reference ``roi_head=dict(...)`` config subtrees, restated as plain data.

These are the exact keys/values of the reference experiment configs
(configs/mv2d/exp/mv2d_r50_frcnn_single_frame_roi_1408x512_ep24.py:40-121 = CFG-S,
configs/mv2d/exp/mv2d_r50_frcnn_two_frames_1408x512_ep24.py:40-125 = CFG-T).
The build's registry (mv2d_amd.registry.build_head) must accept them verbatim.
"""
import copy

POINT_CLOUD_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]   # CFG-T:5
POST_RANGE = [-61.2, -61.2, -10.0, 61.2, 61.2, 10.0]        # CFG-T:6
ROI_SIZE = 7                                                 # CFG-T:7
ROI_STRIDES = [16]                                           # CFG-T:8


def _bbox_head(with_cp):
    return dict(
        type='CrossAttentionBoxHead',
        num_classes=10,
        pc_range=POINT_CLOUD_RANGE,
        transformer=dict(
            type='MV2DTransformer',
            decoder=dict(
                type='PETRTransformerDecoder',
                return_intermediate=True,
                num_layers=6,
                transformerlayers=dict(
                    type='PETRTransformerDecoderLayer',
                    attn_cfgs=[
                        dict(type='FlattenMHSelfAttention', embed_dims=256, num_heads=8, dropout=0.1),
                        dict(type='PETRMultiheadAttention', embed_dims=256, num_heads=8, dropout=0.1),
                    ],
                    feedforward_channels=2048,
                    ffn_dropout=0.1,
                    with_cp=with_cp,
                    operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn', 'norm')),
            )),
        bbox_coder=dict(
            type='NMSFreeCoder',
            post_center_range=POST_RANGE,
            pc_range=POINT_CLOUD_RANGE,
            max_num=300,
            num_classes=10),
        code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.5, 1.5, 2.0, 2.0],
        loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
        loss_bbox=dict(type='L1Loss', loss_weight=0.25),
    )


def _common(with_cp):
    return dict(
        pc_range=POINT_CLOUD_RANGE,
        force_fp32=True,
        bbox_roi_extractor=dict(
            type='SingleRoIExtractor',
            roi_layer=dict(type='RoIAlign', output_size=ROI_SIZE, sampling_ratio=-1),
            featmap_strides=ROI_STRIDES,
            out_channels=512),
        bbox_head=_bbox_head(with_cp),
        query_generator=dict(
            with_avg_pool=True,
            num_shared_convs=1,
            num_shared_fcs=1,
            in_channels=256,
            fc_out_channels=1024,
            roi_feat_size=ROI_SIZE,
            extra_encoding=dict(
                num_layers=2,
                feat_channels=[512, 256],
                features=[dict(type='intrinsic', in_channels=16)]),
        ),
        pe=dict(
            positional_encoding=dict(type='SinePositionalEncoding3D', num_feats=128, normalize=True),
            strides=ROI_STRIDES,
            position_range=POST_RANGE,
            depth_num=64,
            with_fpe=True),
    )


def roi_head_cfg_s():
    """CFG-S:40-121 (MV2D-S single frame)."""
    d = dict(type='MV2DSHead', use_denoise=False)
    d.update(_common(with_cp=False))
    d['box_correlation'] = dict(correlation_mode='topk_matched:1:0.0:0.0')
    return copy.deepcopy(d)


def roi_head_cfg_t():
    """CFG-T:40-125 (MV2D-T two frames)."""
    d = dict(type='MV2DTHead', use_denoise=True, neg_bbox_loss=True,
             denoise_noise_scale=1.25, denoise_split=0.6)
    d.update(_common(with_cp=True))
    d['box_correlation'] = dict(expand_stride=2, correlation_mode='topk_matched:20:0.0:0.0')
    return copy.deepcopy(d)


TEST_CFG_RCNN = dict(score_thr=0.0, nms=dict(nms_thr=1.0, use_rotate_nms=True), max_per_scene=300)  # CFG-T:154-158

# configs/mv2d/exp/mv2d_r50_frcnn_two_frames_1408x512_ep24.py:134-145 (`train_cfg.rcnn`; the single-frame configs carry the same block)
TRAIN_CFG_RCNN = dict(
    stage_loss_weights=[0.1, 0.1, 0.1, 0.1, 0.1, 0.1],
    assigner=dict(type='HungarianAssigner3D', cls_cost=dict(type='FocalLossCost', weight=2.0),
                  reg_cost=dict(type='BBox3DL1Cost', weight=0.25), iou_cost=dict(type='IoUCost', weight=0.0), pc_range=POINT_CLOUD_RANGE),
    sampler_cfg=dict(type='PseudoSampler'), pos_weight=-1, debug=False)
