"""Stub layer, training side: the mmdet==2.25.1 pieces the reference's assigner and loss code call (absent from this image).

TEST INFRASTRUCTURE (see oracle/_stubs.py).  Everything here is a restatement of third-party code from its pinned version — the part of
the training goldens that goes through these classes is "parity unpinned"; the flow around them (HungarianAssigner3D.assign,
normalize_bbox, _get_target_single, loss_single, dn_loss_single, loss) is the reference's own code, imported unmodified.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _stubs
from ._stubs import Registry, _mod


class AssignResult:                                    # mmdet.core.bbox.assigners.AssignResult
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class BaseAssigner:
    pass


class SamplingResult:                                  # mmdet.core.bbox.samplers.SamplingResult
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            if len(gt_bboxes.shape) < 2:
                gt_bboxes = gt_bboxes.view(-1, 4)
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds.long(), :]


class PseudoSampler:                                   # mmdet.core.bbox.samplers.PseudoSampler
    def sample(self, assign_result, bboxes, gt_bboxes, *a, **k):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result)


class FocalLossCost:                                   # mmdet.core.bbox.match_costs.FocalLossCost (binary_input False)
    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12, binary_input=False):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        cls_pred = cls_pred.sigmoid()
        neg_cost = -(1 - cls_pred + self.eps).log() * (1 - self.alpha) * cls_pred.pow(self.gamma)
        pos_cost = -(cls_pred + self.eps).log() * self.alpha * (1 - cls_pred).pow(self.gamma)
        return (pos_cost[:, gt_labels] - neg_cost[:, gt_labels]) * self.weight


class IoUCost:                                         # built by the assigner, never called by its assign()
    def __init__(self, iou_mode='giou', weight=1.):
        self.weight = weight


def _reduce(loss, weight, avg_factor):                 # mmdet.models.losses.utils.weight_reduce_loss, reduction 'mean'
    if weight is not None:
        loss = loss * weight
    return loss.mean() if avg_factor is None else loss.sum() / avg_factor


class FocalLoss(nn.Module):                            # mmdet.models.losses.FocalLoss, use_sigmoid, python path (py_sigmoid_focal_loss)
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0, activated=False):
        super().__init__()
        self.use_sigmoid, self.gamma, self.alpha, self.loss_weight = use_sigmoid, gamma, alpha, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        num_classes = pred.size(1)
        target = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes]
        p = pred.sigmoid()
        target = target.type_as(pred)
        pt = (1 - p) * target + p * (1 - target)
        fw = (self.alpha * target + (1 - self.alpha) * (1 - target)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, target, reduction='none') * fw
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) else weight.view(loss.size(0), -1)
        return self.loss_weight * _reduce(loss, weight, avg_factor)


class L1Loss(nn.Module):                               # mmdet.models.losses.L1Loss
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        if target.numel() == 0:
            return self.loss_weight * pred.sum() * 0
        return self.loss_weight * _reduce(torch.abs(pred - target), weight, avg_factor)


def multi_apply(func, *args, **kwargs):                # mmdet.core.multi_apply
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


class GtBoxes:
    """The two attributes of mmdet3d's LiDARInstance3DBoxes the loss reads (cross_attention_head.py:449-451): bottom-centre boxes
    (x, y, z_bottom, w, l, h, yaw, vx, vy) with `gravity_center` = (x, y, z_bottom + h / 2)."""

    def __init__(self, tensor):
        self.tensor = tensor

    @property
    def gravity_center(self):
        t = self.tensor
        return torch.cat([t[:, :2], t[:, 2:3] + t[:, 5:6] * 0.5], 1)


def install(reference_root='/root/reference'):
    heads = _stubs.install(reference_root)
    import sys
    match_cost = Registry('match_cost')
    assigners = Registry('assigner')
    match_cost.table['FocalLossCost'] = FocalLossCost
    match_cost.table['IoUCost'] = IoUCost
    sys.modules['mmdet.core.bbox.builder'].BBOX_ASSIGNERS = assigners
    _mod('mmdet.core.bbox.assigners', AssignResult=AssignResult, BaseAssigner=BaseAssigner)
    _mod('mmdet.core.bbox.match_costs', build_match_cost=lambda c: match_cost.build(c))
    _mod('mmdet.core.bbox.match_costs.builder', MATCH_COST=match_cost)
    _mod('mmdet.core.bbox.iou_calculators', bbox_overlaps=None)
    root = reference_root.rstrip('/') + '/mmdet3d_plugin'
    for pkg, sub in [('mmdet3d_plugin.core.bbox.assigners', '/core/bbox/assigners'),
                     ('mmdet3d_plugin.core.bbox.match_costs', '/core/bbox/match_costs')]:
        _mod(pkg).__path__ = [root + sub]
    import mmdet3d_plugin.core.bbox.match_costs.match_cost  # noqa: F401  (registers BBox3DL1Cost)
    from mmdet3d_plugin.core.bbox.assigners.hungarian_assigner_3d import HungarianAssigner3D
    import mmdet3d_plugin.models.roi_heads.bbox_heads.cross_attention_head as cah
    cah.multi_apply = multi_apply
    return heads, HungarianAssigner3D


def arm_bbox_head(bbox_head, HungarianAssigner3D, train_cfg, loss_cls, loss_bbox):
    """What CrossAttentionBoxHead.__init__ does with train_cfg / the loss configs when mmdet is present (cross_attention_head.py:110-111,
    150-152)."""
    a = dict(train_cfg['assigner'])
    a.pop('type')
    bbox_head.assigner = HungarianAssigner3D(**a)
    bbox_head.sampler = PseudoSampler()
    bbox_head.loss_cls = FocalLoss(**{k: v for k, v in loss_cls.items() if k != 'type'})
    bbox_head.loss_bbox = L1Loss(**{k: v for k, v in loss_bbox.items() if k != 'type'})
    return bbox_head
