"""Training targets and losses of the box head — host side.

Mirrors the reference interface for this step:

* ``HungarianAssigner3D`` (registry ``BBOX_ASSIGNERS``; mmdet3d_plugin/core/bbox/assigners/hungarian_assigner_3d.py:28-150): the cost
  matrix of all decoder layers is one HIP launch (``mv2d_match_cost``), the assignment itself is scipy's ``linear_sum_assignment`` on
  the host exactly as in the reference (:137), one device->host copy for all layers.
* ``SetPredictionLoss`` (autograd Function) / ``head_loss``: ``CrossAttentionBoxHead.loss`` per layer with the stage weights of
  ``MV2DSHead.forward_train`` (mmdet3d_plugin/models/roi_heads/mv2d_s_head.py:276-303) — sigmoid focal loss + code-weighted L1 on the
  finite targets, forward and gradient in one HIP launch for all layers (``mv2d_set_loss``).
* ``dn_loss``: ``dn_loss_single`` (cross_attention_head.py:477-538) for all layers.

No CPU fallback: the device tensors must be on the GPU and the HIP library present.
"""
import math

import numpy as np
import torch

from . import ops
from .registry import BBOX_ASSIGNERS


def _reduce_mean(v):
    """mmdet.core.reduce_mean on a python scalar: average over the ranks when torch.distributed is up (one all-reduce)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(v)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([float(v)], device=dev)
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()


@BBOX_ASSIGNERS.register_module()
class HungarianAssigner3D:
    def __init__(self, cls_cost=dict(type='FocalLossCost', weight=2.0), reg_cost=dict(type='BBox3DL1Cost', weight=0.25),
                 iou_cost=dict(type='IoUCost', weight=0.0), pc_range=None, solver='native', threads=1):
        if cls_cost.get('type', 'FocalLossCost') != 'FocalLossCost' or reg_cost.get('type', 'BBox3DL1Cost') != 'BBox3DL1Cost':
            raise NotImplementedError('HungarianAssigner3D: only FocalLossCost + BBox3DL1Cost (the shipped configs) are built')
        self.cls_weight = float(cls_cost.get('weight', 1.0))
        self.alpha = float(cls_cost.get('alpha', 0.25))
        self.gamma = float(cls_cost.get('gamma', 2.0))
        self.reg_weight = float(reg_cost.get('weight', 1.0))
        self.pc_range = pc_range
        assert solver in ('native', 'scipy')
        self.solver, self.threads = solver, int(threads)      # threads: host threads of the native solver (1: 0.29 ms for 6 x 300 x 40; SciPy 0.53-0.75)

    def cost(self, bbox_pred, cls_pred, gt_bboxes, gt_labels):
        """[L,R,10], [L,R,C], [G,9] gravity-centre boxes, [G] -> cost [L,R,G] (device)."""
        return ops.match_cost(cls_pred, bbox_pred, gt_bboxes, gt_labels.to(torch.int32), self.cls_weight, self.reg_weight, self.alpha,
                              self.gamma)

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels):
        """Returns match [L,R] int32 (device): index of the assigned ground-truth box or -1 (the reference's ``gt_inds - 1``).
        A 2-D input ([R,10] / [R,C]) is one layer.  The assignment of all layers is ONE host call (``mv2d_lsap_layers``, csrc/lsap.hip: the
        algorithm SciPy's ``linear_sum_assignment`` implements, with its scan order and tie rule -- tests/test_lsap_cpu.py compares them entry by
        entry); ``solver='scipy'`` in the constructor's ``kwargs`` calls SciPy layer by layer as the reference does (:137)."""
        single = bbox_pred.dim() == 2
        if single:
            bbox_pred, cls_pred = bbox_pred[None], cls_pred[None]
        L, R = cls_pred.shape[:2]
        G = gt_bboxes.shape[0]
        match = np.full((L, R), -1, np.int32)
        if R and G:
            cost = self.cost(bbox_pred.contiguous(), cls_pred.contiguous(), gt_bboxes.contiguous(), gt_labels).cpu()     # one device->host copy
            if self.solver == 'scipy':
                from scipy.optimize import linear_sum_assignment
                cost = cost.numpy()
                for l in range(L):
                    rows, cols = linear_sum_assignment(cost[l])
                    match[l, rows] = cols
            else:
                from ._lib import check, load
                check(load().mv2d_lsap_layers(cost.data_ptr(), L, R, G, match.ctypes.data, self.threads), 'mv2d_lsap_layers')
        out = torch.from_numpy(match).to(cls_pred.device)
        return out[0] if single else out


class SetPredictionLoss(torch.autograd.Function):
    """Returns the per-layer losses times the layer weights, [L,2] = (layer_w[l] * loss_cls[l], layer_w[l] * loss_bbox[l]) — the entries of
    the reference's loss dict — differentiable in the logits (through the first column only) and the box codes (second column only):
    forward and both gradients are one launch (``mv2d_set_loss``)."""

    @staticmethod
    def forward(ctx, cls, box, match, gt, gt_labels, code_w, layer_w, cls_avg, box_avg, alpha, gamma, w_cls, w_box, skip_bg):
        need = cls.requires_grad or box.requires_grad
        loss, dcls, dbox = ops.set_loss(cls.contiguous(), box.contiguous(), match, gt, gt_labels, code_w, layer_w, cls_avg, box_avg, alpha,
                                        gamma, w_cls, w_box, skip_bg, need_grad=need)
        ctx.save_for_backward(dcls, dbox)
        return loss * layer_w[:, None]

    @staticmethod
    def backward(ctx, g):
        dcls, dbox = ctx.saved_tensors                      # gradients of layer_w[l] * loss_cls[l] / layer_w[l] * loss_bbox[l]
        return (dcls * g[:, 0, None, None], dbox * g[:, 1, None, None]) + (None,) * 12


class HeadLoss:
    """The loss part of ``CrossAttentionBoxHead`` (constructor arguments as in the reference config: ``loss_cls``, ``loss_bbox``,
    ``code_weights``, ``train_cfg['assigner']``)."""

    def __init__(self, num_classes=10, loss_cls=None, loss_bbox=None, code_weights=None, train_cfg=None, device='cuda'):
        loss_cls = loss_cls or dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0)
        loss_bbox = loss_bbox or dict(type='L1Loss', loss_weight=0.25)
        if loss_cls.get('type') != 'FocalLoss' or not loss_cls.get('use_sigmoid', True) or loss_bbox.get('type') != 'L1Loss':
            raise NotImplementedError('HeadLoss: sigmoid FocalLoss + L1Loss (the shipped configs) only')
        self.num_classes = num_classes
        self.alpha, self.gamma = float(loss_cls.get('alpha', 0.25)), float(loss_cls.get('gamma', 2.0))
        self.w_cls, self.w_box = float(loss_cls.get('loss_weight', 1.0)), float(loss_bbox.get('loss_weight', 1.0))
        cw = code_weights or [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2]
        self.code_weights = torch.tensor(cw[:10], dtype=torch.float32, device=device)
        dn = list(cw[:10])
        dn[6] = dn[7] = 0.0                                              # cross_attention_head.py:531
        self.dn_code_weights = torch.tensor(dn, dtype=torch.float32, device=device)
        train_cfg = train_cfg or {}
        a = dict(train_cfg.get('assigner') or dict(type='HungarianAssigner3D'))
        self.assigner = BBOX_ASSIGNERS.build(a)
        self.stage_loss_weights = train_cfg.get('stage_loss_weights')
        self.device = device

    def _layer_w(self, L, scale=1.0):
        # (constants: uploaded once -- a tensor built from a Python list is a blocking host-to-device copy in every step)
        key = (L, float(scale))
        cache = self.__dict__.setdefault('_lw', {})
        if key not in cache:
            w = self.stage_loss_weights or [1.0] * L
            cache[key] = torch.tensor([float(x) * scale for x in w[:L]], dtype=torch.float32, device=self.device)
        return cache[key]

    def loss(self, all_cls_scores, all_bbox_preds, gt_bboxes, gt_labels, match=None):
        """all_cls_scores [L,R,C], all_bbox_preds [L,R,10] (one sample), gt_bboxes [G,9] gravity-centre boxes, gt_labels [G].
        Returns (losses, total): the reference's dict keys ``l{i}.loss_cls`` / ``l{i}.loss_bbox`` (already times the stage weight,
        mv2d_s_head.py:299-302), every entry differentiable (``sum(losses.values()).backward()`` as mmdet's ``_parse_losses`` does), their sum,
        and the assignment."""
        L, R = all_cls_scores.shape[:2]
        gt_bboxes = gt_bboxes.to(self.device, torch.float32).contiguous()
        labels32 = gt_labels.to(self.device, torch.int32).contiguous()
        if match is None:
            match = self.assigner.assign(all_bbox_preds.detach(), all_cls_scores.detach(), gt_bboxes, labels32)
        num_pos = min(R, gt_bboxes.shape[0])                             # a full assignment matches min(R, G) pairs in every layer
        cls_avg = max(num_pos * 1.0, 1.0)                                # bg_cls_weight = 0, sync_cls_avg_factor False
        box_avg = max(_reduce_mean(num_pos), 1.0)                        # reduce_mean(num_total_pos).clamp(min=1), :419-420
        lw = self._layer_w(L)
        weighted = SetPredictionLoss.apply(all_cls_scores, all_bbox_preds, match, gt_bboxes, labels32, self.code_weights, lw,
                                           cls_avg, box_avg, self.alpha, self.gamma, self.w_cls, self.w_box, False)
        total = weighted.sum()
        vals = weighted.reshape(-1).unbind(0)              # one autograd node for the 2 L entries (a select per entry is 2 L zero-filled gradients)
        losses = {}
        for l in range(L):
            losses[f'l{l}.loss_cls'] = vals[2 * l]
            losses[f'l{l}.loss_bbox'] = vals[2 * l + 1]
        return losses, total, match

    def dn_loss(self, output_known_class, output_known_coord, known_bboxs, known_labels, num_tgt, split, neg_bbox_loss=False,
                denoise_weight=1.0):
        """``dn_loss_single`` for all layers: outputs of the denoising queries [L,N,C] / [L,N,10], their targets known_bboxs [N,9],
        known_labels [N] (== num_classes: negative).  Keys ``l{i}.dn_loss_cls`` / ``l{i}.dn_loss_bbox`` (mv2d_s_head.py:292-297)."""
        L, N = output_known_class.shape[:2]
        cls_avg = max(num_tgt * 3.14159 / 6 * split * split * split, 1.0)
        box_avg = max(_reduce_mean(num_tgt), 1.0)
        lw = self._layer_w(L, denoise_weight)
        match = torch.arange(N, dtype=torch.int32, device=self.device).repeat(L, 1)
        weighted = SetPredictionLoss.apply(output_known_class, output_known_coord, match,
                                           known_bboxs.to(self.device, torch.float32).contiguous(),
                                           known_labels.to(self.device, torch.int32).contiguous(), self.dn_code_weights, lw,
                                           cls_avg, box_avg, self.alpha, self.gamma, self.w_cls, self.w_box, not neg_bbox_loss)
        total = weighted.sum()
        vals = weighted.reshape(-1).unbind(0)
        losses = {}
        for l in range(L):
            losses[f'l{l}.dn_loss_cls'] = vals[2 * l]
            losses[f'l{l}.dn_loss_bbox'] = vals[2 * l + 1]
        return losses, total


def prepare_for_dn(reference_points, gt_bboxes, gt_labels, denoise_scalar=10, denoise_noise_scale=1.0, denoise_noise_trans=0.0,
                   denoise_split=0.75, num_classes=10, pc_range=None, rnd=None, eps=1e-4, dense_mask=True):
    """``MV2DSHead.prepare_for_dn`` for the training branch and one sample (mmdet3d_plugin/models/roi_heads/mv2d_s_head.py:39-121;
    ``batch_size`` is 1 there, :249).  reference_points [R,3] (device), gt_bboxes [G,9] gravity-centre boxes, gt_labels [G];
    ``rnd`` [G*denoise_scalar,3] uniform in [0,1) stands for the reference's ``torch.rand_like`` (drawn on the device when None).

    Returns (padded_reference_points [1, pad+R, 3], attn_mask bool [pad+R, pad+R] (True = masked) or None when ``dense_mask`` is False,
    mask_dict).  mask_dict has the reference's keys plus ``dn_single``: the attention kernels take (pad_size, dn_single) instead of the
    dense mask (``ops.self_attn_dn``)."""
    dev = reference_points.device
    G = int(gt_bboxes.shape[0])
    R = int(reference_points.shape[0])
    pad = G * denoise_scalar
    gt = gt_bboxes.to(dev, torch.float32).contiguous()
    lab = gt_labels.to(dev, torch.int32).contiguous()
    if rnd is None and denoise_noise_scale > 0:
        rnd = torch.rand(pad, 3, device=dev)
    ref, known_labels, known_bboxs = ops.dn_queries(gt, lab, rnd, denoise_scalar, denoise_noise_scale, denoise_noise_trans, denoise_split,
                                                    num_classes, pc_range or [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], eps)
    padded = torch.cat([ref, reference_points.to(torch.float32)], 0)[None]
    attn_mask = None
    if dense_mask:
        i = torch.arange(pad + R, device=dev)
        grp = i // max(G, 1)
        visible = (i[None, :] >= pad) | ((i[:, None] < pad) & (grp[:, None] == grp[None, :]))
        attn_mask = ~visible
    idx = torch.arange(G, device=dev)
    mask_dict = {
        'known_indice': idx.repeat(denoise_scalar),
        'batch_idx': torch.zeros(G, dtype=torch.long, device=dev),
        'map_known_indice': torch.arange(pad, device=dev),
        'known_lbs_bboxes': (known_labels, known_bboxs),
        'know_idx': [torch.ones(G, dtype=torch.long, device=dev)],
        'pad_size': pad,
        'dn_single': G,
    }
    return padded, attn_mask, mask_dict


def fallback_key_csr(row_ptr, col_idx, key_index):
    """Training-time rule of the two-frame head (RH/mv2d_t_head.py:80-82): a RoI that sees no key at all gets the key at map position
    (view 0, 0, 0) un-masked instead of a fully masked row (NaN).  row_ptr [R+1] / col_idx [nnz] int32 -> the CSR with one entry
    ``key_index`` (the compacted index of that position) in every empty row; also returns the bool mask of the patched rows."""
    R = row_ptr.numel() - 1
    counts = (row_ptr[1:] - row_ptr[:-1]).long()
    empty = counts == 0
    if not bool(empty.any()):
        return row_ptr, col_idx, empty
    counts2 = torch.where(empty, torch.ones_like(counts), counts)
    rp = torch.zeros(R + 1, dtype=torch.int32, device=row_ptr.device)
    rp[1:] = counts2.cumsum(0).to(torch.int32)
    col = torch.full((int(rp[-1].item()),), int(key_index), dtype=torch.int32, device=row_ptr.device)
    before = empty.long().cumsum(0) - empty.long()                    # patched rows in front of every row
    row_of = torch.repeat_interleave(torch.arange(R, device=row_ptr.device), counts)
    col[torch.arange(col_idx.numel(), device=row_ptr.device) + before[row_of]] = col_idx
    return rp, col, empty


class TrainDecoder:
    """Differentiable decoder + prediction heads of ``CrossAttentionBoxHead`` for training: the dense linears, layer
    norms and the FFN go through ``mv2d_amd.autograd_ops`` (forward and backward on the HIP split-precision GEMM / layer-norm kernels; torch
    keeps the autograd graph and the element-wise glue), **both attentions through the HIP sparse-attention kernels**
    (``ops.SparseCrossAttention``: forward ``mv2d_sparse_xattn_fwd``, backward ``mv2d_sparse_xattn_bwd``) — the self attention as a CSR
    with the denoising mask of ``prepare_for_dn`` as its pattern.  Parameters are read from the head module itself (the reference's
    state-dict names), so their ``.grad`` is what an optimizer / DDP sees.  K / V are rounded to bf16 for the attention kernels (as in
    the inference engine); everything else is fp32.

    Mirrors PETRTransformerDecoder (6 post-norm layers: self_attn, norm, cross_attn, norm, ffn, norm; shared post_norm on every
    intermediate; mmdet3d_plugin/models/utils/petr_transformer.py) and the branches of cross_attention_head.py:200-242."""

    def __init__(self, roi_head, num_heads=8):
        self.p = dict(roi_head.named_parameters())
        bh = roi_head.bbox_head
        self.L, self.pc_range, self.H = bh.num_pred, [float(x) for x in roi_head.pc_range], num_heads
        # dropout (shipped configs: 0.1 on both attentions and in the FFN, configs/mv2d/exp/*:67-79): the module tree carries the
        # nn.Dropout objects of the reference's layers (mmcv MultiheadAttention: attn_drop on the probabilities + dropout_layer on the
        # output path; FFN: after the activation and after the second linear); they are applied here when the head is in training mode
        self.roi_head = roi_head
        self.layers = getattr(getattr(bh.transformer, 'decoder', None), 'layers', None)
        self._warned = False
        import os
        self.fused = os.environ.get('MV2D_TRAIN_FUSED', '1') != '0'      # the C-issued decoder (round 5); '0': the per-operator autograd graph
        self.batched_dense = True          # per-operator graph: the denoising rows' dense block batched over the heads (False: rounds 3-4's loop)

    def _drops(self, i, j):
        """(attention-probability p, output-path callable) of attention j of layer i; identity / 0 in eval mode or without a module tree"""
        if self.layers is None or not self.roi_head.training:
            return 0.0, (lambda t: t)
        att = self.layers[i].attentions[j]
        p_attn = float(getattr(getattr(att, 'attn', None), 'dropout', 0.0) or 0.0)
        pd, dl = getattr(att, 'proj_drop', None), getattr(att, 'dropout_layer', None)
        return p_attn, (lambda t: (dl(pd(t) if pd is not None else t) if dl is not None else t))

    def _next_seed(self, p_attn):
        """seed of the next attention-probability dropout mask: a counter on top of torch's seed (torch.manual_seed makes a run repeatable; no
        device synchronisation)"""
        if p_attn <= 0:
            return 0
        import torch.distributed as dist
        self._drop_calls = getattr(self, '_drop_calls', 0) + 1
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0      # data-parallel ranks draw different masks
        return (torch.initial_seed() * 2654435761 + self._drop_calls * 40503 + rank * 7919) & 0xffffffff

    @staticmethod
    def self_attention_pattern(T, pad, single, device):
        """CSR of prepare_for_dn's attn_mask: key j visible to query i iff j >= pad, or i < pad and i // single == j // single."""
        i = torch.arange(T, device=device)
        grp = i // max(single, 1)
        vis = (i[None, :] >= pad) | ((i[:, None] < pad) & (grp[:, None] == grp[None, :]))
        row_ptr = torch.zeros(T + 1, dtype=torch.int32, device=device)
        row_ptr[1:] = vis.sum(1).cumsum(0).to(torch.int32)
        return row_ptr, vis.nonzero()[:, 1].to(torch.int32).contiguous()

    def _attn(self, q_in, k_in, v_in, name, csr, tr, dense=None, p_attn=0.0):
        """dense = (n, keys): the first n query rows see exactly the key rows `keys` (the denoising rows of the cross attention) — a dense
        block, computed with batched GEMMs instead of n rows of len(keys) pairs each in the sparse kernels; csr then covers rows n.."""
        import torch.nn.functional as F
        from .autograd_ops import in_proj, linear, matmul_nt_ad
        w, b = self.p[name + '.attn.in_proj_weight'], self.p[name + '.attn.in_proj_bias']
        Cc = q_in.shape[-1]
        # (round 3: every projection forward + backward on the HIP GEMM, mv2d_amd/autograd_ops.py; no rocBLAS)
        q, k, v = in_proj(q_in, k_in, v_in, w, b, 1.0 / (Cc // self.H) ** 0.5)       # one autograd node: dW / db land in one [3C, C] / [3C] buffer
        if dense is not None and dense[0] > 0:
            n, keys = dense
            kd, vd = k[keys], v[keys]
            if self.batched_dense:
                from .autograd_ops import DenseHeadsAttnFn
                tops = [DenseHeadsAttnFn.apply(q[:n], kd, vd, self.H, p_attn)]       # all heads: one batched launch per product (round 5)
            else:
                H, d = self.H, Cc // self.H
                tops = []
                for h in range(H):                  # the dense denoising block per head: logits and P V as HIP products, softmax / dropout element-wise
                    sl = slice(h * d, (h + 1) * d)
                    prob = F.dropout(torch.softmax(matmul_nt_ad(q[:n, sl].contiguous(), kd[:, sl].contiguous()), -1), p_attn, p_attn > 0)
                    tops.append(matmul_nt_ad(prob, vd[:, sl].t().contiguous()))
            ctx = torch.cat([torch.cat(tops, 1), ops.SparseCrossAttention.apply(q[n:], k, v, csr[0], csr[1], False, tr, p_attn, self._next_seed(p_attn))])
        else:
            ctx = ops.SparseCrossAttention.apply(q, k, v, csr[0], csr[1], False, tr, p_attn, self._next_seed(p_attn))
        return linear(ctx, self.p[name + '.attn.out_proj.weight'], self.p[name + '.attn.out_proj.bias'])

    def __call__(self, ref, key_in, val_in, row_ptr, col_idx, pad=0, single=1, dt=0.0, dn_keys=None):
        """ref [T,3] normalised reference points (denoising rows first), key_in / val_in [S,256] (memory + key_pos, memory), CSR of the
        cross attention over the T rows — or, with ``dn_keys`` (sorted key rows every denoising query sees), over the T - pad matched
        rows only.  Returns (all_cls [L,T,C], all_reg [L,T,10]); rows >= pad get v / dt when dt != 0."""
        import torch.nn.functional as F
        from .autograd_ops import layer_norm, linear
        P, T, dev = self.p, ref.shape[0], ref.device
        pre = 'bbox_head.transformer.decoder.'
        ref = ref.to(torch.float32)
        if getattr(self, '_dim_t', None) is None or self._dim_t.device != dev:
            dim_t = torch.arange(128, dtype=torch.float32)
            self._dim_t = (10000 ** (2 * (dim_t // 2) / 128)).to(dev).contiguous()          # MU/pe.py:24-25, as mv2d_amd.calib builds it
        dim_t = self._dim_t
        if ref.requires_grad:                                                           # gradient into the query generator
            from .autograd_ops import PosEmbFn
            posemb = PosEmbFn.apply(ref, dim_t)
        else:
            posemb = ops.posemb3d(ref.contiguous(), dim_t)
        qpos = linear(linear(posemb, P['bbox_head.query_embedding.0.weight'], P['bbox_head.query_embedding.0.bias'], 1),
                      P['bbox_head.query_embedding.2.weight'], P['bbox_head.query_embedding.2.bias'])
        key_in, val_in = key_in.float(), val_in.float()
        S = key_in.shape[0]
        sa, sa_t = self._sa_pattern(T, pad, single, dev)
        ca = (row_ptr.contiguous(), col_idx.contiguous())
        use_c = self.fused and T > 0 and S > 0        # (an empty sample takes the per-operator graph: nothing to launch)
        # the pattern grouped by key is only read by the backward: the C route builds it there (the backward has host time to spare)
        ca_t = None if use_c else ops.csr_transpose(ca[0], ca[1], S)
        ln = lambda t, n: layer_norm(t, P[n + '.weight'], P[n + '.bias'])  # noqa: E731      (mv2d_row_ln / mv2d_layer_norm_bwd)
        training = self.layers is not None and self.roi_head.training
        if use_c:
            # (round 5) the six layers + post_norm as one autograd node, launch sequences issued from C (csrc/train_decoder.hip)
            outs = self._fused_layers(qpos, key_in, val_in, sa, sa_t, ca, ca_t, training, pad if dn_keys is not None else 0, dn_keys)
            return self._branches(outs, ref, pad, dt)
        x = torch.zeros(T, qpos.shape[1], device=dev)
        outs = []
        for i in range(self.L):
            lp = f'{pre}layers.{i}.'
            p_sa, drop_sa = self._drops(i, 0)
            p_ca, drop_ca = self._drops(i, 1)
            # (round 3: the attention-probability dropout of both attentions runs inside the sparse kernels, mv2d_sparse_xattn_*_drop)
            x = ln(x + drop_sa(self._attn(x + qpos, x + qpos, x, lp + 'attentions.0', sa, sa_t, p_attn=p_sa)), lp + 'norms.0')
            x = ln(x + drop_ca(self._attn(x + qpos, key_in, val_in, lp + 'attentions.1', ca, ca_t,
                                          None if dn_keys is None else (pad, dn_keys.long()), p_attn=p_ca)), lp + 'norms.1')
            h = linear(x, P[lp + 'ffns.0.layers.0.0.weight'], P[lp + 'ffns.0.layers.0.0.bias'], 1)
            y = None
            if training:
                ffn = self.layers[i].ffns[0]                      # mmcv FFN: Linear-ReLU-Dropout, Linear-Dropout, (+ dropout_layer) + identity
                h = ffn.layers[0][2](h)
            y = linear(h, P[lp + 'ffns.0.layers.1.weight'], P[lp + 'ffns.0.layers.1.bias'])
            if training:
                y = ffn.layers[2](y)
                dl = getattr(ffn, 'dropout_layer', None)
                y = dl(y) if dl is not None else y
            x = ln(x + y, lp + 'norms.2')
            outs.append(ln(x, pre + 'post_norm'))
        # (with denoising queries the layers stay on the per-operator graph -- their dense block is not in the C entry -- the branches need not)
        return self._branches(torch.stack(outs) if self.fused else outs, ref, pad, dt)

    def _sa_pattern(self, T, pad, single, dev):
        """the self-attention pattern and its transpose: a function of (T, pad, single) only, kept between steps"""
        key = (T, pad, single, str(dev))
        if getattr(self, '_sa_key', None) != key:
            sa = self.self_attention_pattern(T, pad, single, dev)
            self._sa_key, self._sa = key, (sa, ops.csr_transpose(sa[0], sa[1], T))
        return self._sa

    def _fused_layers(self, qpos, key_in, val_in, sa, sa_t, ca, ca_t, training, pad=0, dn_keys=None):
        from .autograd_ops import DECODER_PARAMS, DecoderFn
        P, pre = self.p, 'bbox_head.transformer.decoder.'
        params = [P[f'{pre}layers.{i}.{n}'] for i in range(self.L) for n in DECODER_PARAMS] + [P[pre + 'post_norm.weight'], P[pre + 'post_norm.bias']]
        drops = (0.0,) * 6
        if training:
            lay = self.layers[0]                      # (the shipped configs build every layer from one dict)
            comb = lambda *ps: 1.0 - math.prod(1.0 - float(p) for p in ps)  # noqa: E731   two dropouts in a row = one with the product of the keep rates
            pdrop = lambda m: float(getattr(m, 'p', 0.0) or 0.0)  # noqa: E731

            def att(a):
                return float(getattr(a.attn, 'dropout', 0.0) or 0.0), comb(pdrop(getattr(a, 'proj_drop', None)), pdrop(getattr(a, 'dropout_layer', None)))
            ffn = lay.ffns[0]
            drops = att(lay.attentions[0]) + att(lay.attentions[1]) + (pdrop(ffn.layers[0][2]), comb(pdrop(ffn.layers[2]), pdrop(getattr(ffn, 'dropout_layer', None))))
        meta = dict(L=self.L, sa=sa, sa_t=sa_t, ca=ca, ca_t=ca_t, drops=drops, seed=self._next_seed(1.0 if any(drops) else 0.0), pad=pad,
                    dn_keys=None if (dn_keys is None or pad == 0) else dn_keys.to(torch.int32).contiguous())
        return DecoderFn.apply(qpos.contiguous(), key_in.contiguous(), val_in.contiguous(), meta, *[p if p.is_contiguous() else p.contiguous() for p in params])

    def _branches(self, outs, ref, pad, dt):
        """classification / regression branches and the box code of every intermediate output (cross_attention_head.py:200-242)"""
        import torch.nn.functional as F
        from .autograd_ops import layer_norm, linear
        P, dev = self.p, ref.device
        ln = lambda t, n: layer_norm(t, P[n + '.weight'], P[n + '.bias'])  # noqa: E731
        if torch.is_tensor(outs) and outs.shape[1] == 0:
            outs = list(outs.unbind(0))
        if torch.is_tensor(outs):
            # (round 5) all branches of all layers as one autograd node, the layers side by side on streams (mv2d_train_heads_fwd / _bwd)
            from .autograd_ops import BRANCH_PARAMS, HeadsFn
            bp = [P['bbox_head.' + n.format(l=l)] for l in range(self.L) for n in BRANCH_PARAMS]
            all_cls_t, t = HeadsFn.apply(outs, *[p if p.is_contiguous() else p.contiguous() for p in bp])
            return all_cls_t, self._box_code(t, ref, pad, dt)
        all_cls, ts = [], []
        for l in range(self.L):
            c, g = f'bbox_head.cls_branches.{l}.', f'bbox_head.reg_branches.{l}.'
            y = F.relu(ln(linear(outs[l], P[c + '0.weight'], P[c + '0.bias']), c + '1'))
            y = F.relu(ln(linear(y, P[c + '3.weight'], P[c + '3.bias']), c + '4'))
            all_cls.append(linear(y, P[c + '6.weight'], P[c + '6.bias']))
            t = linear(outs[l], P[g + '0.weight'], P[g + '0.bias'], 1)
            ts.append(linear(linear(t, P[g + '2.weight'], P[g + '2.bias'], 1), P[g + '4.weight'], P[g + '4.bias']))
        return torch.stack(all_cls), self._box_code(torch.stack(ts), ref, pad, dt)

    def _box_code(self, t, ref, pad, dt):
        """raw code [L,T,10] -> boxes (cross_attention_head.py:219-233): one launch per direction (autograd_ops.BoxCodeFn)"""
        from .autograd_ops import BoxCodeFn
        return BoxCodeFn.apply(t, ref, self.pc_range, pad, float(dt or 0.0))


def allreduce_gradients(parameters, bucket_bytes=256 << 20, average=True, group=None):
    """Data-parallel gradient synchronisation of a training step ("DDP all-reduce over RCCL"): the gradients are packed
    into flat fp32 buckets and each bucket is ONE all-reduce (backend ``nccl`` = RCCL on ROCm).  The whole head has 5.6 M parameters =
    22 MB, so with the default bucket size a step is a single collective — xGMI rings are per-link latency-bound for small messages, one
    large message beats the reference's DDP default of 25 MB buckets per ~100 tensors.  Parameters without a gradient on some rank (e.g.
    no denoising rows on a sample without ground truth) take part with zeros, so that every rank issues the same collectives.
    Returns the number of all-reduces issued; a no-op (0) without an initialised process group or with one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    params = [p for p in parameters if p.requires_grad]
    world = dist.get_world_size(group)
    calls, i = 0, 0
    while i < len(params):
        bucket, size = [], 0
        while i < len(params) and (not bucket or size + params[i].numel() * 4 <= bucket_bytes):
            bucket.append(params[i])
            size += params[i].numel() * 4
            i += 1
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in bucket])
        dist.all_reduce(flat, group=group)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        calls += 1
    return calls


def query_generator_autograd(roi_head, roi_feat, intr_feat, minv):
    """``QueryGenerator.forward`` + ``center2lidar`` + the reference-point normalisation (RH/utils/query_generator.py:352-405,333-341;
    RH/mv2d_s_head.py:146-152) as torch autograd over the module's own parameters, on the engine's RoIAlign output: roi_feat [R,49,256]
    (bf16, cell-major 7x7), intr_feat [R,16] (scaled intrinsics), minv [R,16] = inverse(K_roi E^T) fp32.  Returns the normalised
    reference points [R,3], differentiable w.r.t. the query generator's parameters and w.r.t. roi_feat (``ops.RoIAlignRows`` carries the
    gradient on to the feature map)."""
    import torch.nn.functional as F
    from .autograd_ops import linear
    qg = roi_head.query_generator
    R = roi_feat.shape[0]
    Cc = roi_feat.shape[-1]
    # conv3x3 (padding 1) as im2col + ONE product on the HIP GEMM, k order (tap, channel); the unfolding is one launch per direction (round 5)
    from .autograd_ops import Im2Col3x3Fn
    assert Cc == 256
    cols = Im2Col3x3Fn.apply(roi_feat.float().reshape(R, 49, Cc))
    conv = qg.shared_convs[0].conv
    x = linear(cols, conv.weight.permute(0, 2, 3, 1).reshape(conv.weight.shape[0], 9 * Cc), conv.bias, 1).view(R, 49, -1).mean(1)   # ReLU, AvgPool2d(7)
    x = linear(x, qg.shared_fcs[0].weight, qg.shared_fcs[0].bias, 1)
    x = torch.cat([x, intr_feat.detach().float()], 1).clamp(min=-5e3, max=5e3)
    x = linear(x, qg.extra_enc[0].weight, qg.extra_enc[0].bias, 1)
    x = linear(x, qg.extra_enc[2].weight, qg.extra_enc[2].bias, 1)
    c = linear(x, qg.fc_center.weight, qg.fc_center.bias)
    from .autograd_ops import Center2LidarFn
    return Center2LidarFn.apply(c, minv.detach().reshape(R, 16), [float(v) for v in roi_head.pc_range])      # one launch per direction (round 5)


def key_embedding_autograd(roi_head, A1, A2, Xf):
    """The PE block at the gathered key positions (MU/pe.py:36-48,150-166) as torch autograd over the module's parameters, on the inputs the
    engine prepared: A1 [S,192] inverse-sigmoid frustum coordinates, A2 [S,384] sine embedding (bf16, no gradient), Xf [S,256] feature rows
    (differentiable).  Returns (feat + pe, feat, pe) [S,256] fp32: the T path's keys / values; the S path RoI-aligns pe."""
    from .autograd_ops import linear
    pe = roi_head.position_encoding
    lin = lambda conv, t, act=0: linear(t, conv.weight.flatten(1), conv.bias, act)  # noqa: E731   (1x1 convs on the HIP GEMM)
    feat = Xf.float()
    p3d = lin(pe.position_encoder[2], lin(pe.position_encoder[0], A1.detach().float(), 1))
    gate = torch.sigmoid(lin(pe.fpe.conv_expand, lin(pe.fpe.conv_reduce, feat, 1)))
    sine = lin(pe.adapt_pos3d[2], lin(pe.adapt_pos3d[0], A2.detach().float(), 1))
    pos = p3d * gate + sine
    return feat + pos, feat, pos
