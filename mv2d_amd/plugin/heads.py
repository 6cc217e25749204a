"""MV2DHead / MV2DSHead / MV2DTHead / CrossAttentionBoxHead — the reference's RoI-head registry surface
(RH/mv2d_head.py, RH/mv2d_s_head.py, RH/mv2d_t_head.py, RH/bbox_heads/cross_attention_head.py) on MI355X.

``simple_test(x, proposal_list, img_metas, rescale=False)`` keeps the reference signature and return value
(``[[boxes, scores, labels]]``) and runs the fused HIP engine (mv2d_amd.engine.HeadEngine), which is built lazily
from the module's own ``state_dict`` — so a checkpoint loaded with the reference key layout is what runs.
``forward_train`` keeps the reference signature and loss dict; with gradients enabled it runs the autograd route (HIP attention / RoIAlign /
loss kernels under torch autograd for the dense part) and fills the gradients of every parameter of the head and of the feature map
(LOG.md 7.1).
"""
import copy

import numpy as np
import torch
import torch.nn as nn

from .. import calib, ops
from ..engine import HeadEngine
from ..registry import HEADS, build_bbox_coder, build_head, build_loss, build_roi_extractor, build_transformer
from .modules import BoxCorrelation, PE, QueryGenerator, _f, _rows

C = 256


@HEADS.register_module()
class CrossAttentionBoxHead(nn.Module):
    """RH/bbox_heads/cross_attention_head.py:86-242,357-377 (forward + get_bboxes)."""

    def __init__(self, num_classes, transformer, pc_range, embed_dims=256, num_reg_fcs=2, group_reg_dims=(2, 2, 1, 1, 2, 2),
                 use_reg_layer=False, pre_embed=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0),
                 bbox_coder=dict(type='NMSFreeCoder', post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                 pc_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], max_num=100, num_classes=10),
                 sync_cls_avg_factor=False, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        assert not use_reg_layer and not pre_embed and embed_dims == C and num_reg_fcs == 2 and num_classes == 10, \
            'kernel path implements the shipped head configuration'
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.transformer = build_transformer(transformer)
        self.pc_range, self.embed_dims, self.pre_embed = pc_range, embed_dims, pre_embed
        self.query_embedding = nn.Sequential(nn.Linear(embed_dims * 3 // 2, embed_dims), nn.ReLU(), nn.Linear(embed_dims, embed_dims))
        self.num_pred = transformer['decoder']['num_layers']
        self.num_classes = self.cls_out_channels = num_classes
        cls_branch = []
        for _ in range(num_reg_fcs):
            cls_branch += [nn.Linear(embed_dims, embed_dims), nn.LayerNorm(embed_dims), nn.ReLU(inplace=True)]
        cls_branch.append(nn.Linear(embed_dims, num_classes))
        reg_branch = []
        for _ in range(num_reg_fcs):
            reg_branch += [nn.Linear(embed_dims, embed_dims), nn.ReLU()]
        reg_branch.append(nn.Linear(embed_dims, sum(group_reg_dims)))
        self.cls_branches = nn.ModuleList([copy.deepcopy(nn.Sequential(*cls_branch)) for _ in range(self.num_pred)])
        self.reg_branches = nn.ModuleList([copy.deepcopy(nn.Sequential(*reg_branch)) for _ in range(self.num_pred)])
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.code_size = kwargs.get('code_size', 10)
        cw = kwargs.get('code_weights', [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2])[:self.code_size]
        self.code_weights = nn.Parameter(torch.tensor(cw, requires_grad=False), requires_grad=False)
        self.sync_cls_avg_factor, self.train_cfg, self.test_cfg = sync_cls_avg_factor, train_cfg, test_cfg
        self.bg_cls_weight = 0
        self.fp16_enabled = False

    def init_weights(self):
        self.transformer.init_weights()
        for m in self.cls_branches:
            nn.init.constant_(m[-1].bias, float(-np.log((1 - 0.01) / 0.01)))

    def position_embedding(self, query_pos):
        shp = query_pos.shape[:-1]
        ct = calib.constant_tables()
        pe = ops.posemb3d(query_pos.reshape(-1, 3).float().contiguous(), ct['dim_t'].to(query_pos.device))
        q0, q2 = self.query_embedding[0], self.query_embedding[2]
        h = ops.gemm_f32(pe, _f(q0.weight), _f(q0.bias), act=1)
        return ops.gemm_f32(h, _f(q2.weight), _f(q2.bias)).view(*shp, C)

    def forward(self, reference_points, x, masks, pos_embed, attn_mask=None, cross_attn_mask=None, force_fp32=False,
                query_embeds=None, return_query_feats=False, **kwargs):
        """reference_points [bs,Q,3], x / pos_embed [bs,n,c,h,w], masks [bs,n,h,w] -> (all_cls_scores, all_bbox_preds) [L,bs,Q,10].
        Stray kwargs (e.g. ``pe=(module, x, metas)`` of RH/mv2d_head.py:172) are accepted and ignored like in the reference."""
        assert not self.training, 'inference kernels only (SURVEY.md §8 f3)'
        if not self.pre_embed:
            query_embeds = self.position_embedding(reference_points)
        outs_dec, _ = self.transformer(x.float(), masks, query_embeds.float(), pos_embed.float(), attn_mask=attn_mask,
                                       cross_attn_mask=cross_attn_mask)
        L, bs, Q, _ = outs_dec.shape
        M = bs * Q
        dev = outs_dec.device
        od = outs_dec.reshape(L, M, C).float().contiguous()
        st = lambda idx, attr: torch.stack([_f(getattr(br[idx], attr)) for br in self._cur]).contiguous()
        gk = dict(groups=L, a_gs=M * C, c_gs=M * C)
        self._cur = self.cls_branches
        h1 = ops.gemm_f32(od, st(0, 'weight'), st(0, 'bias'), M=M, lda=C, ldc=C, **gk)
        h2 = ops.row_ln(h1.view(L * M, C), ln=(st(1, 'weight'), st(1, 'bias')), relu=True, rows_per_group=M).view(L, M, C)
        h1 = ops.gemm_f32(h2, st(3, 'weight'), st(3, 'bias'), M=M, lda=C, ldc=C, **gk)
        h2 = ops.row_ln(h1.view(L * M, C), ln=(st(4, 'weight'), st(4, 'bias')), relu=True, rows_per_group=M).view(L, M, C)
        cls = torch.empty((L, M, 10), device=dev)
        ops.gemm_f32(h2, st(6, 'weight'), st(6, 'bias'), out=cls, M=M, lda=C, ldc=10, groups=L, a_gs=M * C, c_gs=M * 10)
        self._cur = self.reg_branches
        r1 = ops.gemm_f32(od, st(0, 'weight'), st(0, 'bias'), act=1, M=M, lda=C, ldc=C, **gk)
        r2 = ops.gemm_f32(r1, st(2, 'weight'), st(2, 'bias'), act=1, M=M, lda=C, ldc=C, **gk)
        reg = torch.empty((L, M, 10), device=dev)
        ops.gemm_f32(r2, st(4, 'weight'), st(4, 'bias'), out=reg, M=M, lda=C, ldc=10, groups=L, a_gs=M * C, c_gs=M * 10)
        del self._cur
        ops.finalize_reg(reg, reference_points.reshape(M, 3).float().contiguous(), L, M, torch.tensor(self.pc_range, dtype=torch.float32), 0.0)
        all_cls_scores, all_bbox_preds = cls.view(L, bs, Q, 10), reg.view(L, bs, Q, 10)
        if return_query_feats:
            return all_cls_scores, all_bbox_preds, outs_dec[-1]
        return all_cls_scores, all_bbox_preds

    def get_bboxes(self, preds_dicts, img_metas, rescale=False):
        preds = self.bbox_coder.decode(preds_dicts)
        ret = []
        for i, p in enumerate(preds):
            bboxes = p['bboxes']
            bboxes[:, 2] = bboxes[:, 2] - bboxes[:, 5] * 0.5
            box_type = img_metas[i].get('box_type_3d') if isinstance(img_metas[i], dict) else None
            if box_type is not None:
                bboxes = box_type(bboxes, bboxes.size(-1))
            ret.append([bboxes, p['scores'], p['labels']])
        return ret

    def _hip_loss(self, device):
        # the HIP loss kernels behind the reference's per-layer interface: no stage weights here, the RoI head applies them (mv2d_s_head.py:292-302)
        from ..train import HeadLoss
        key = (str(device), self.code_weights.data_ptr(), self.code_weights._version)
        if getattr(self, '_hl_key', None) != key:
            lc = dict(self.loss_cls.cfg, type=getattr(self.loss_cls, 'type', 'FocalLoss'), use_sigmoid=self.loss_cls.use_sigmoid)
            lb = dict(self.loss_bbox.cfg, type=getattr(self.loss_bbox, 'type', 'L1Loss'))
            tc = dict(assigner=(self.train_cfg or {}).get('assigner')) if self.train_cfg else None
            self._hl = HeadLoss(num_classes=self.num_classes, loss_cls=lc, loss_bbox=lb, code_weights=[float(x) for x in self.code_weights],
                                train_cfg=tc, device=device)
            self._hl_key = key
        return self._hl

    def loss(self, gt_bboxes_3d_list, gt_labels_3d_list, preds_dicts, cls_reg_targets=None, gt_bboxes_ignore=None):
        """cross_attention_head.py:436-463: Hungarian assignment + sigmoid focal loss + code-weighted L1 of ONE decoder layer
        (``preds_dicts['cls_scores']`` [1,R,C], ``['bbox_preds']`` [1,R,10]) -> ``dict(loss_cls, loss_bbox)``, both differentiable, through
        mv2d_match_cost / scipy / mv2d_set_loss (train.HeadLoss).  One sample per call like the RoI head (mv2d_head.py:251);
        precomputed ``cls_reg_targets`` are not supported (the reference's RoI heads never pass them)."""
        assert gt_bboxes_ignore is None, f'{self.__class__.__name__} only supports for gt_bboxes_ignore setting to None.'
        if cls_reg_targets is not None:
            raise NotImplementedError('CrossAttentionBoxHead.loss: precomputed cls_reg_targets')
        cls_scores, bbox_preds = preds_dicts['cls_scores'], preds_dicts['bbox_preds']
        assert cls_scores.shape[0] == 1 and len(gt_bboxes_3d_list) == 1, 'one sample per call'
        dev = cls_scores.device
        g = gt_bboxes_3d_list[0]
        gt = g if torch.is_tensor(g) else torch.cat((g.gravity_center, g.tensor[:, 3:]), dim=1)
        losses, _, _ = self._hip_loss(dev).loss(cls_scores.float().contiguous(), bbox_preds.float().contiguous(),
                                               gt.to(dev, torch.float32).contiguous(), gt_labels_3d_list[0].to(dev))
        return dict(loss_cls=losses['l0.loss_cls'], loss_bbox=losses['l0.loss_bbox'])

    def dn_loss_single(self, cls_scores, bbox_preds, known_bboxs, known_labels, num_total_pos, pc_range=None, split=0.75, neg_bbox_loss=False):
        """cross_attention_head.py:476-538 for one layer: outputs of the denoising queries [N,C] / [N,10] (or with a leading 1) against their
        targets -> (dn_loss_cls, dn_loss_bbox) (the reference's return order, weighted by its ``dn_weight`` = 1)."""
        dev = cls_scores.device
        c = cls_scores.reshape(1, -1, cls_scores.shape[-1]).float().contiguous()
        b = bbox_preds.reshape(1, -1, bbox_preds.shape[-1]).float().contiguous()
        losses, _ = self._hip_loss(dev).dn_loss(c, b, known_bboxs, known_labels, num_total_pos, split, neg_bbox_loss=neg_bbox_loss)
        return losses['l0.dn_loss_cls'], losses['l0.dn_loss_bbox']


@HEADS.register_module()
class MV2DHead(nn.Module):
    """RH/mv2d_head.py:18-267 (base RoI head: T-path forward with use_denoise never set)."""

    KIND = 'T'

    def __init__(self, bbox_roi_extractor, bbox_head, query_generator, pe, box_correlation, pc_range, intrins_feat_scale=0.1,
                 feat_lvl=0, force_fp32=False, train_cfg=None, test_cfg=None, **kwargs):
        super().__init__()
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_roi_extractor = build_roi_extractor(bbox_roi_extractor)
        bbox_head = dict(bbox_head)
        bbox_head.update(dict(train_cfg=train_cfg, test_cfg=test_cfg))
        self.bbox_head = build_head(bbox_head)
        self.roi_size = bbox_roi_extractor['roi_layer']['output_size']
        if isinstance(self.roi_size, int):
            self.roi_size = [self.roi_size, self.roi_size]
        query_generator = dict(query_generator)
        query_generator.update(dict(loss_cls=self.bbox_head.loss_cls))
        self.query_generator = QueryGenerator(**query_generator)
        self.position_encoding = PE(**pe)
        self.box_corr_module = BoxCorrelation(**box_correlation)
        self.pc_range, self.intrins_feat_scale, self.feat_lvl, self.force_fp32 = pc_range, intrins_feat_scale, feat_lvl, force_fp32
        self.stage_loss_weights = train_cfg.get('stage_loss_weights') if train_cfg else None
        self._engine, self._engine_ver = None, None

    with_bbox = True

    @property
    def strides(self):
        return self.position_encoding.strides

    @property
    def num_classes(self):
        return self.bbox_head.num_classes

    # ---- engine plumbing ------------------------------------------------------------------------------
    def _engine_num_views(self, img_metas):
        return len(img_metas)

    def engine(self, device, img_metas, allow_stale=False):
        """The fused engine packed from the module's current parameters (re-packed whenever a parameter changed).  ``allow_stale``: the
        caller only reads what does not depend on the parameters (RoI list, correlation / key list / CSR, PE inputs, per-RoI cameras) — the
        autograd route of forward_train —, so an engine packed from older parameter values will do and an optimizer step does not force
        a re-pack."""
        tail = (str(device), self._engine_num_views(img_metas))
        if allow_stale and self._engine is not None and self._engine_ver[-2:] == tail:
            return self._engine
        ver = tuple((p.data_ptr(), p._version) for p in self.parameters()) + tail
        if self._engine is None or self._engine_ver != ver:
            sd = {k: v for k, v in self.state_dict().items()}
            bc = self.box_corr_module
            coder = self.bbox_head.bbox_coder
            self._engine = HeadEngine(sd, self.KIND, device, num_views=self._engine_num_views(img_metas), topk=bc.topk,
                                      expand_stride=bc.expand_stride, num_layers=self.bbox_head.num_pred, max_num=coder.max_num,
                                      pc_range=tuple(self.pc_range), post_range=tuple(coder.post_center_range),
                                      depth_num=self.position_encoding.depth_num, stride=self.strides[self.feat_lvl],
                                      iou_thr=bc.iou_thr, ratio=bc.ratio, num_classes=self.bbox_head.num_classes,
                                      masked_row=(self.test_cfg or {}).get('masked_row', 'nan'),
                                      exact=(self.test_cfg or {}).get('index_exact', None))      # None: MV2D_EXACT decides
            if 'lo8_rows' in (self.test_cfg or {}):                              # test_cfg.lo8_rows=False: fp16 lo halves of the key / value rows (engine.py; default: e4m3 bytes)
                self._engine.lo8_rows = bool(self.test_cfg['lo8_rows'])
            self._engine_ver = ver
        return self._engine

    def simple_test(self, x, proposal_list, img_metas, rescale=False):
        """x: list with the stride-16 map [V,256,h,w]; proposal_list: V x [n,6]; img_metas: V dicts -> [[boxes, scores, labels]]."""
        assert len(img_metas) // img_metas[0]['num_views'] == 1
        feat = x[self.feat_lvl]
        eng = self.engine(feat.device, img_metas)
        self.box_corr_module.check_counts([len(p) for p in proposal_list])
        out = eng.run(feat.float(), proposal_list, img_metas)
        boxes, scores, labels = eng.results(out)
        boxes = boxes.clone()
        box_type = img_metas[0].get('box_type_3d')
        if box_type is not None:
            boxes = box_type(boxes, boxes.size(-1))
        return [[boxes, scores.clone(), labels.clone()]]

    def simple_test_batch(self, x, proposal_lists, img_metas_list, rescale=False):
        """Several samples through ONE sequence of launches (the reference asserts one sample per call, mv2d_head.py / SURVEY.md 2.2;
        a dataloader with samples_per_gpu > 1 would call this once instead of looping over simple_test).
        x: list with the stacked stride-16 map [B*V,256,h,w] (sample-major, as the backbone / neck of a batch produce it);
        proposal_lists: B x (V x [n,6]); img_metas_list: B x (V dicts) -> B x [boxes, scores, labels]."""
        B = len(proposal_lists)
        feat = x[self.feat_lvl]
        assert feat.shape[0] % B == 0 and all(len(m) == feat.shape[0] // B for m in img_metas_list)
        eng = self.engine(feat.device, img_metas_list[0])
        out = eng.run_batch(feat.float(), proposal_lists, img_metas_list)
        res = []
        for b, (boxes, scores, labels) in enumerate(eng.results_batch(out)):
            boxes = boxes.clone()
            box_type = img_metas_list[b][0].get('box_type_3d')
            if box_type is not None:
                boxes = box_type(boxes, boxes.size(-1))
            res.append([boxes, scores.clone(), labels.clone()])
        return res

    def _forward_train_autograd(self, eng, out, hl, gt, labels, dn_noise, feat):
        from .. import train
        ws, R = out['ws'], out['R']
        # everything that enters an autograd Function is COPIED out of the engine's workspace: the next run() on the same bucket rewrites
        # those buffers through raw pointers (a second forward before backward(), an eval hook), which autograd's version counters cannot see
        row_ptr = ws['row_ptr'][:R + 1].clone()
        # the input map, position-major, as a differentiable view: its gradient comes back through RoIAlign and the key rows
        V, _, h, w = feat.shape
        fm = feat.float().permute(0, 2, 3, 1).reshape(V * h * w, C)
        rois = ws['rois'][:R].clone()
        bbox_feats = ops.RoIAlignRows.apply(fm, None, rois, h, w)                                   # [R,49,256]
        # reference points with the gradient of the query generator (issued before the first host read-back below: the host keeps
        # launching while the engine's kernels run)
        ref = train.query_generator_autograd(self, bbox_feats, ws['enc'][:R, 1024:1040].clone(), ws['minv'][:R].clone())
        # ONE read-back for the data-dependent sizes: allowed pairs, listed positions, (T) whether a RoI has no key, (T) the key of position 0
        empty = (row_ptr[1:] == row_ptr[:-1]).any().to(torch.int32) if self.KIND == 'T' else row_ptr.new_zeros(())
        nnz, S, any_empty, s0 = (int(v) for v in torch.stack([row_ptr[R], ws['S_dev'].reshape(()).to(torch.int32), empty, ws['pos2s'][0].to(torch.int32)]).tolist())
        col = ws['col_idx'][:nnz].clone()
        A1, A2, s2pos = ws['A1'][:S].clone(), ws['A2'][:S].clone(), ws['s2pos'][:S].long()
        if self.KIND == 'T' and any_empty:
            # a RoI without a single visible key: in training the reference un-masks the key at map position (view 0, 0, 0) for it
            # (RH/mv2d_t_head.py:80-82) instead of producing a NaN row; that position joins the key list if no RoI lists it
            if s0 < 0:
                a1, a2 = eng.pe_input_rows(ws, torch.zeros(1, dtype=torch.int32, device=fm.device), V, h, w)
                A1, A2, s2pos, s0 = torch.cat([A1, a1]), torch.cat([A2, a2]), torch.cat([s2pos, s2pos.new_zeros(1)]), S
            row_ptr, col, _ = train.fallback_key_csr(row_ptr, col, s0)
        # the PE block at the positions the engine listed (T: the gathered keys; S: every position a RoIAlign tap can touch)
        key_rows, val_rows, pe_rows = train.key_embedding_autograd(self, A1, A2, fm[s2pos])
        if self.KIND == 'T':
            key_in, val_in = key_rows, val_rows
        else:
            pe_aligned = ops.RoIAlignRows.apply(pe_rows, ws['pos2s'].clone(), rois, h, w, fm.detach())
            val_in = bbox_feats.reshape(R * 49, C)
            key_in = val_in + pe_aligned.reshape(R * 49, C)
        ref_const, pad, single, md, keys = ws['ref'][:R].clone(), 0, 1, None, None
        if getattr(self, 'use_denoise', False):
            padded, _, md = train.prepare_for_dn(ref_const, gt, labels, self.denoise_scalar, self.denoise_noise_scale, self.denoise_noise_trans,
                                                 self.denoise_split, self.num_classes, list(self.pc_range), rnd=dn_noise, dense_mask=False)
            pad, single = md['pad_size'], max(md['dn_single'], 1)
            ref = torch.cat([padded[0, :pad], ref])
            keys = torch.unique(col)                   # the denoising rows see every key some RoI can see: a dense block
        if getattr(self, '_train_decoder', None) is None:
            self._train_decoder = train.TrainDecoder(self)
        all_cls, all_reg = self._train_decoder(ref, key_in, val_in, row_ptr, col, pad, single, float(out.get('dt', 0.0)),
                                               dn_keys=keys if pad > 0 else None)
        losses = {}
        if pad > 0:
            known_labels, known_bboxs = md['known_lbs_bboxes']
            dn, _ = hl.dn_loss(all_cls[:, :pad], all_reg[:, :pad], known_bboxs, known_labels, pad, self.denoise_split,
                               neg_bbox_loss=self.neg_bbox_loss, denoise_weight=self.denoise_weight)
            losses.update(dn)
        main, _, _ = hl.loss(all_cls[:, pad:], all_reg[:, pad:], gt, labels)
        losses.update(main)
        return losses

    def _head_loss(self, device):
        from ..train import HeadLoss
        bh = self.bbox_head
        cw_key = (bh.code_weights.data_ptr(), bh.code_weights._version)          # a checkpoint loaded later changes the code weights
        if getattr(self, '_hl', None) is None or self._hl.device != device or getattr(self, '_hl_key', None) != cw_key:
            self._hl_key = cw_key
            # the CONFIGURED loss types are handed on, so that an unsupported one raises in HeadLoss instead of training with focal + L1
            lc = dict(bh.loss_cls.cfg, type=getattr(bh.loss_cls, 'type', 'FocalLoss'), use_sigmoid=bh.loss_cls.use_sigmoid)
            lb = dict(bh.loss_bbox.cfg, type=getattr(bh.loss_bbox, 'type', 'L1Loss'))
            self._hl = HeadLoss(num_classes=bh.num_classes, loss_cls=lc, loss_bbox=lb, code_weights=[float(x) for x in bh.code_weights],
                                train_cfg=self.train_cfg, device=device)
        return self._hl

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_3d, gt_labels_3d, ori_gt_bboxes_3d,
                      ori_gt_labels_3d, attr_labels=None, gt_bboxes_ignore=None, gt_masks=None, dn_noise=None, autograd=None, **kwargs):
        """The reference's signature (RH/mv2d_head.py:196-246, RH/mv2d_s_head.py:235-305) and loss dict (keys ``l{i}.loss_cls`` /
        ``l{i}.loss_bbox`` / ``l{i}.dn_loss_cls`` / ``l{i}.dn_loss_bbox``, times the stage weights).  Two routes (LOG.md 7.1):
        ``autograd=False`` — everything through the fused engine kernels, forward only; ``autograd=True`` (default when gradients are
        enabled) — query generator / PE / key gathering through the engine (no gradient: they are treated as constants), decoder + heads
        through ``train.TrainDecoder`` (torch autograd around the HIP attention forward / backward kernels) and the HIP loss kernel, so
        ``sum(losses.values()).backward()`` fills the gradients of the decoder, the branches and ``query_embedding``.
        ``ori_gt_bboxes_3d[0]``: a LiDARInstance3DBoxes-like object (``gravity_center``, ``tensor``) or a [G,9] tensor of gravity-centre
        boxes; ``dn_noise`` [G*denoise_scalar,3] in [0,1) replaces the on-device draw of the denoising noise."""
        from .. import train
        assert len(img_metas) // img_metas[0]['num_views'] == 1
        feat = x[self.feat_lvl]
        dev = feat.device
        if autograd is None:
            autograd = torch.is_grad_enabled() and any(p.requires_grad for p in self.bbox_head.parameters())
        eng = self.engine(dev, img_metas, allow_stale=bool(autograd))
        eng.keep_sine_rows = bool(autograd)                 # the autograd route evaluates the sine branch of the PE block itself
        eng.stop_before_decoder = bool(autograd)            # ... and runs its own decoder: the engine stops after the query generator
        try:
            out = eng.run(feat.detach().float(), [p[:, :6] for p in proposal_list], img_metas)
        finally:
            eng.stop_before_decoder = False
        g = ori_gt_bboxes_3d[0]
        gt = g if torch.is_tensor(g) else torch.cat((g.gravity_center, g.tensor[:, 3:]), dim=1)
        gt = gt.to(dev, torch.float32).contiguous()
        labels = ori_gt_labels_3d[0].to(dev)
        hl = self._head_loss(dev)
        R = out['R']
        losses = {}
        if autograd:
            return self._forward_train_autograd(eng, out, hl, gt, labels, dn_noise, feat)
        if getattr(self, 'use_denoise', False):
            ref = out['ws']['ref'][:R]
            padded, _, md = train.prepare_for_dn(ref, gt, labels, self.denoise_scalar, self.denoise_noise_scale, self.denoise_noise_trans,
                                                 self.denoise_split, self.num_classes, list(self.pc_range), rnd=dn_noise, dense_mask=False)
            pad = md['pad_size']
            all_cls, all_reg = eng.train_forward(out, padded[0, :pad], md['dn_single'])
            if pad > 0:
                known_labels, known_bboxs = md['known_lbs_bboxes']
                dn, _ = hl.dn_loss(all_cls[:, :pad].contiguous(), all_reg[:, :pad].contiguous(), known_bboxs, known_labels, pad,
                                   self.denoise_split, neg_bbox_loss=self.neg_bbox_loss, denoise_weight=self.denoise_weight)
                losses.update(dn)
            all_cls, all_reg = all_cls[:, pad:].contiguous(), all_reg[:, pad:].contiguous()
        else:
            all_cls, all_reg = eng.train_forward(out)
        main, _, _ = hl.loss(all_cls, all_reg, gt, labels)
        losses.update(main)
        return losses


@HEADS.register_module()
class MV2DSHead(MV2DHead):
    """RH/mv2d_s_head.py:18-305 (eval branch :181-192: RoI-gather cross attention)."""

    KIND = 'S'

    def __init__(self, use_denoise=False, neg_bbox_loss=False, denoise_scalar=10, denoise_noise_scale=1.0, denoise_noise_trans=0.0,
                 denoise_weight=1.0, denoise_split=0.75, **kwargs):
        super().__init__(**kwargs)
        self.use_denoise, self.neg_bbox_loss, self.denoise_scalar = use_denoise, neg_bbox_loss, denoise_scalar
        self.denoise_noise_scale, self.denoise_noise_trans = denoise_noise_scale, denoise_noise_trans
        self.denoise_weight, self.denoise_split = denoise_weight, denoise_split


@HEADS.register_module()
class MV2DTHead(MV2DSHead):
    """RH/mv2d_t_head.py:18-142 (masked-map cross attention, velocity / dt for two-frame input)."""

    KIND = 'T'

    def __init__(self, num_views=6, **kwargs):
        super().__init__(**kwargs)
        self.num_views = num_views

    def _engine_num_views(self, img_metas):
        return self.num_views
