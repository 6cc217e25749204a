"""The extra FPN level MV2D puts between the 2-D detector and the RoI head ("next" row f2 of SURVEY.md §8(f)).

Config (configs/mv2d/exp/*:32-39): ``neck=dict(type='FPN', in_channels=[256]*5, out_channels=256, start_level=2, end_level=2,
num_outs=1)`` — mmdet's FPN restricted to ONE level: ``out = fpn_conv3x3(lateral_conv1x1(inputs[2]))``, plain convolutions with
bias (ConvModule with norm_cfg=None, act_cfg=None); called from ``MV2D.process_detector_feat``
(mmdet3d_plugin/models/detectors/mv2d.py:122-127).  mmdet==2.25.1 is third-party and absent from the reference tree: parity-unpinned,
restated in oracle.fpn_neck.  State-dict keys follow mmdet's FPN (``lateral_convs.0.conv.*``, ``fpn_convs.0.conv.*``).

The output is returned as a [V,256,h,w] tensor in channels_last memory format: exactly the position-major fp32 map the RoI-head engine
reads, so no transpose runs between the neck and the head.  bf16 MFMA with fp32 accumulation (both convolutions).
"""
import torch
import torch.nn as nn

from .. import ops
from ..registry import NECKS

C = 256


class _Conv(nn.Module):
    """ConvModule(norm_cfg=None, act_cfg=None) -> state-dict key ``conv.weight`` / ``conv.bias``."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2)


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False, relu_before_extra_convs=False,
                 no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'), init_cfg=None):
        super().__init__()
        if end_level in (-1, len(in_channels) - 1) and len(in_channels) - start_level != 1:
            raise NotImplementedError('only the single-level FPN of the MV2D configs (start_level == end_level, num_outs == 1) is built')
        end = len(in_channels) - 1 if end_level == -1 else end_level
        if end != start_level or num_outs != 1 or add_extra_convs or norm_cfg is not None or act_cfg is not None:
            raise NotImplementedError('only the single-level FPN of the MV2D configs (start_level == end_level, num_outs == 1) is built')
        if in_channels[start_level] != C or out_channels != C:
            raise NotImplementedError('256 -> 256 channels only')
        self.level = start_level
        self.lateral_convs = nn.ModuleList([_Conv(C, C, 1)])
        self.fpn_convs = nn.ModuleList([_Conv(C, C, 3)])
        self._packed, self._ver = None, None

    def _weights(self):
        lat, fpn = self.lateral_convs[0].conv, self.fpn_convs[0].conv
        ver = (lat.weight.data_ptr(), lat.weight._version, fpn.weight.data_ptr(), fpn.weight._version, str(lat.weight.device))
        if self._ver != ver:
            w1 = ops.f32_to_bf16(lat.weight.detach().float().reshape(C, C).contiguous())
            w3 = ops.f32_to_bf16(fpn.weight.detach().float().permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous())   # [out][tap][cin]
            self._packed = (w1, lat.bias.detach().float().contiguous(), ops.pack_wfrag(w3), fpn.bias.detach().float().contiguous())
            self._ver = ver
        return self._packed

    def forward(self, inputs):
        x = inputs[self.level]
        assert x.is_cuda and x.dim() == 4 and x.shape[1] == C
        V, _, h, w = x.shape
        w1, b1, w3p, b3 = self._weights()
        xcl = ops.nchw_to_nhwc_bf16(x.float().contiguous())                          # [P,256] bf16
        lat = ops.gemm_bf16(xcl, w1, b1)                                             # 1x1 lateral conv, bf16 out
        out = ops.map_conv3x3(lat, w3p, b3, V, h, w)                                 # [P,256] fp32 position-major
        return (out.view(V, h, w, C).permute(0, 3, 1, 2),)                           # NCHW view, channels_last memory
