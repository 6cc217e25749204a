"""Fused MI355X inference engine for the MV2D RoI head hot path.

One ``HeadEngine`` owns the packed weights (fp16 hi / lo pairs, fragment-major, for the split-precision kernels of the key and the query side;
single fp16 copies for the query generator's conv), a shape-keyed workspace (every buffer pre-allocated, kernels never allocate) and enqueues the whole
frame on the current HIP stream through the C-ABI (mv2d_amd.ops) without a single device->host synchronisation:

  T path (MV2DTHead, RH/mv2d_t_head.py:26-142)        S path (MV2DSHead eval branch, RH/mv2d_s_head.py:122-211)
  rois -> per-RoI camera -> RoIAlign(feat) -> QueryGenerator -> ref points -> query_pos
  box correlation -> match lists                        box correlation (k=1) -> CSR over RoI-feature rows
  masks -> key list S + CSR (device-side counts)        RoI tap positions -> PE there -> RoIAlign(feat, pe)
  PE only at the S key positions -> key rows (feat + pe) / value rows (feat), key16, shared by all layers and heads
  6 x [self-attn, LN, cross-attn in the raw key space (K / V in_proj folded into the query side), LN, FFN, LN] -> heads -> top-k decode

The reference evaluates PE on the whole map and runs dense [8,R,S] attention with a boolean mask.  The engine's route is the INDEX-EXACT one
(round 5: the default): every operand an fp16 hi + lo pair, three MFMAs per product -- the ranked box indices equal the reference's (DESIGN.md
section 2).  ``exact=False`` selects the opt-in key16 mode (one fp16 rounding of the key side: faster, a few ranked indices move).
"""
import math
import os

import numpy as np
import torch

from . import calib, ops

BF16 = torch.bfloat16
F32 = torch.float32
C = 256
L_DEFAULT = 6


def _t(x, device, dtype=None):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    x = x.detach()
    if dtype is not None:
        x = x.to(dtype)
    return x.to(device).contiguous()


class HeadEngine:
    def __init__(self, state_dict, kind, device, num_views=6, topk=None, expand_stride=None, num_layers=L_DEFAULT,
                 max_num=300, pc_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0),
                 post_range=(-61.2, -61.2, -10.0, 61.2, 61.2, 10.0), depth_num=64, stride=16, col_cap_per_query=2048,
                 iou_thr=0.0, ratio=0.0, masked_row='nan', exact=None, num_classes=10):
        assert kind in ('S', 'T')
        self.kind = kind
        self.dev = torch.device(device)
        self.L = num_layers
        self.num_views = num_views
        self.topk = topk if topk is not None else (1 if kind == 'S' else 20)
        self.expand = float(expand_stride if expand_stride is not None else (0 if kind == 'S' else 2))
        self.max_num = max_num
        self.num_classes = int(num_classes)
        assert self.num_classes == 10, 'the fused prediction-branch kernels (mv2d_heads_fused*) are built for 10 classes'
        self.depth_num = depth_num
        self.stride = stride
        self.iou_thr, self.ratio = iou_thr, ratio
        self.col_cap_per_query = col_cap_per_query
        # a query whose every key is masked: 'nan' = the reference's behaviour (nn.MultiheadAttention yields NaN, the next self
        # attention spreads it to every query, the frame returns no boxes), 'zero' = zero attention output for that query only
        assert masked_row in ('nan', 'zero')
        self.empty_nan = masked_row == 'nan'
        self.pc_range_h = torch.tensor(pc_range, dtype=F32)
        self.post_range_h = torch.tensor(post_range, dtype=F32)
        self.post_range_h64 = torch.tensor(post_range, dtype=torch.float64)
        self.const = {k: v.to(self.dev) for k, v in calib.constant_tables().items()}
        self._ws = {}
        self._ws_base = {}
        self._shape_cache = {}
        self.prof = None              # dict name -> [events] when stage timing is on (bench.py)
        self.fork_qg = True           # T path: query-generator chain on a second stream
        # ---- route options (round 4: the retired A/B routes left the engine; what remains are ATTRIBUTES a caller may set before the first
        # run, every one of them part of the hipGraph key -- DESIGN.md "switches" table).  The only environment variable the engine reads is
        # MV2D_EXACT.
        # adapt_pos3d(sine) (MU/pe.py:164-166) depends only on the weights and on the padding geometry of the rig, not on features, boxes
        # or calibration: it is constant-folded into a per-(weights version, geometry) table that the fused PE kernel adds in its epilogue
        self._weights_version = 0     # bumped by every load_state(): invalidates whatever was folded from the weights
        self.stop_before_decoder = False   # the autograd training route only needs geometry, RoI features, PE inputs and reference points of a run
        self.keep_sine_rows = False   # the training route reads the per-key sine rows (ws['A2']) although the inference kernel does not
        self.force_nc = None          # bench only (S path): overwrite the correlation lists so that every query reads n_c RoIs
        # tests only (eager runs): keep the pre-softmax per-head logits of every layer's cross attention (out['stages']['dbg_logits']
        # [L,8,col_cap] in CSR order, WITHOUT the per-(query, head) constant q_h . bk_h that cancels in the softmax) and the scaled,
        # projected queries ('dbg_q' [L,R,256]) they were computed from
        self.debug_attn = False
        # The per-head query / context maps of the tile cross attention run inside the neighbouring row kernels (mv2d_attn_out_qmap_x3 /
        # _zmap_x3: 6 instead of 8 launches per layer, bitwise the same results) for SMALL launches (<= 512 query rows, i.e. one sample per
        # call: the two saved launches per layer count there) and as separate kernels for batches (a row kernel is bound by streaming its
        # weights through ONE CU per 32 rows; the fused ones stream twice as much).  None: by the row count; True / False forces it.
        self.fuse_maps = {'0': False, '1': True}.get(os.environ.get('MV2D_FUSE_MAPS', ''), None)      # (the variable: A/B runs)
        # OPT-IN: evaluate the cls / reg branches of the last decoder layer only (what decoding reads).  Not the default: out['cls'] / out['reg']
        # then carry stale rows for the other layers, and the reference's forward does evaluate all six.
        self.last_stage_heads = False
        # T path: the blocks of the per-query tile kernel run in the order of the queries' SMALLEST KEY (mv2d_xattn_query_order): neighbouring
        # blocks of an XCD then read overlapping key sets from its L2 (cfg3_t 54.8 -> 47.6 us per layer; bitwise the same results)
        # S path (round 4): the queries ranked by the smallest RoI they list (own or matched; computed from the correlation lists inside the
        # launch that builds the CSR) -- matched RoIs of different views then run side by side; a no-op for queries that read only their own RoI
        self.q_order = True
        # waves per query of the tile kernel: the kernel alone takes the same time with 1, 2 or 4 (it is bound by what the memory system
        # delivers), but a launch with fewer waves leaves more of the chip to the other streams' kernels: cfg2_s 8067 / 8043 / 7869
        # samples/s for 1 / 2 / 4, cfg3_t (rows of ~200 keys) 5803 / 5868 / 5758
        self.xattn_waves = int(os.environ.get('MV2D_XATTN_WAVES', '2'))       # (the variable: A/B runs)
        # S path (rows of similar length): query map -> tile attention -> context map as ONE launch per layer (csrc/xattn_fused.hip, round 5: blocks
        # of 8 queries, Qt / z stay on chip; bitwise the three kernels with one wave per query).  None: on the S path when no debug output is asked
        # for; False / True forces it.  In the graph key.
        self.fuse_xattn = None
        # Round 6, OPT-IN: key tiles SHARED between the queries of a group (csrc/xattn_group.hip: one block per 8 queries that are neighbours in the
        # launch order walks the union of their key lists once through an LDS-DMA ring; wave = head, query / context maps in the same launch).  It
        # reads what it should (1.34-1.68 x the distinct rows instead of 2.99 x at cfg3_t) and is SLOWER than the per-query kernels (242-326 us
        # against 158 + 33 us per cfg3_t layer): eight heads x every union tile is 2.7 x the (wave, tile) steps of the per-query walk, and the
        # 32 KB tiles of the hi + lo route leave the 160 KB of LDS no room to run the queries' own walks side by side (LOG.md, round 6).  None /
        # False: off; True forces it.  In the graph key.
        self.group_xattn = None
        # Layer 0 of the decoder starts from target = 0 (RH/bbox_heads/cross_attention_head.py:32): the VALUE rows of its self attention are
        # in_proj_v(0) + b_v = b_v for every query, the softmax weights of a row sum to 1, so its context is b_v whatever the queries are
        # (MU/petr_transformer.py:317-370: value = key before the positional embedding = target).  The engine feeds rows of b_v to the
        # out-projection kernel instead of launching the in-projection and the attention core of layer 0 (the reference's own sum of
        # probabilities is 1 +- 1e-7; a NaN query position still poisons the frame one layer later, through its cross attention).  In the
        # graph key; the training forward (denoising mask) keeps the launches.
        self.fold_sa0 = True
        # transpose only the map rows inside some RoI's rectangle (see _enqueue); in the graph key
        self.masked_transpose = True
        # INDEX-EXACT ROUTE = THE DEFAULT since round 5 (exact=None -> True; exact=False / MV2D_EXACT=0 / test_cfg.index_exact=False selects the
        # opt-in "key16" mode with ONE fp16 rounding of the key side: ~1.3 x faster, 4-22 of 300 ranked indices differ from the reference's).
        # Every 16-bit rounding of the key side is replaced by fp32-class arithmetic -- the PE block in one split-precision kernel on unrounded inputs (csrc/pe_x3.hip), the key / value
        # rows of the tile attention (and, without exact_skip={'conv'}, the query generator's conv) as key16 hi + lo pairs -- so that the INTEGER outputs (labels, bbox_index)
        # can be compared bit for bit with the reference's (tests/test_gpu_golden.py).  Enqueue-only and hipGraph-replayable like the
        # default route (bench.py: samples_s_index_exact).
        self.exact = (os.environ.get('MV2D_EXACT', '1') != '0') if exact is None else bool(exact)
        # Stages of the index-exact route that run with the default route's single key16 rounding -- any of 'attn' (hi rows only in the tile
        # attention), 'pe' (fused key16 PE kernel), 'conv' (single-precision RoI conv); in the graph key.  Round 4 (tools/ablate_exact.py,
        # profiles/r04_ablate_exact.txt): with fp16 cells the query generator's conv in SINGLE precision leaves the ranked indices and the
        # class-logit error of the index-exact route where they are (cfg2_s 2 -> 2, cfg3_t 4 -> 4, cfg5_t 2 -> 0 of 300; cls 5.0e-6 -> 6.2e-6 /
        # 5.1e-6 -> 4.9e-6 / 5.8e-6 -> 5.7e-6): its 2304-term dot products average the 2^-12 roundings down, and the 3 x MFMA-bound split-
        # precision kernel (362 vs 123 us per 2400 RoIs) leaves the route.  Attention rows and PE stay hi + lo: dropping either costs ranks.
        self.exact_skip = frozenset({'conv'})
        # Round 6: the lo halves of the key / value rows as 8-BIT floats (csrc/common.h "lo8": OCP e4m3 of lo * 2^12, 256-byte rows) -- a (query, key)
        # pair of the cross attention gathers 1.5 KB instead of 2 KB and the row producers write a quarter less.  The lo part carries 2^-12 of a
        # product, its e4m3 rounding 2^-16: all 17 reference parity cases keep their ranked indices (class logits 6.2e-7 .. 1.1e-6 of their range
        # against 6.3e-7 .. 8.0e-7 with key16 lo rows, bound 3e-6; an e5m2 lo half or a missing one moves ranks: profiles/r06_ablate_exact.txt).
        # The kernels decode the bytes to key16 in registers: results are bitwise those of key16 lo rows holding the decoded values.  False: key16 lo
        # rows (rounds 3-5).  The shared-tile kernel (group_xattn) reads key16 lo rows only.  In the graph key.
        self.lo8_rows = True
        # Round 6, S path: the PE rows are written AT THEIR MAP POSITIONS (a position-indexed fp32 map, zero-filled once) instead of compacted in key-list
        # order, so that RoIAlign reads its second map without the position -> row table: one dependent load less in front of every bilinear tap of the
        # PE map (the kernel is a chain of such round trips).  Same values, same arithmetic.  Off for keep_stages runs (they expose the
        # compact rows) and in key16 mode; MV2D_PE_AT_POS=0 turns it off (A/B runs).  In the graph key.
        self.pe_at_positions = os.environ.get('MV2D_PE_AT_POS', '1') != '0'
        # experiments only (tools/ablate_exact.py; needs lo8_rows = False): zero the lo halves of the value / key rows after they were written -- what a route with hi-only value
        # (or key) rows would compute, at the full route's cost
        self.ablate_zero_lo = frozenset()
        # Round 6, OPT-IN: the split-precision PE block on the second shape of its kernel (csrc/pe_x3b.hip: a wave owns 32 rows through both layers of
        # each MLP, the hidden layer stays in registers, the weights go through an LDS-DMA ring shared by the block's 4 waves; BITWISE the outputs of
        # csrc/pe_x3.hip).  Measured no faster (1213 vs 1165-1244 us per 250 k rows; 8 waves x 16 rows: 1056 us, bound by 8 x 32 KB of LDS reads per
        # k-step): LOG.md round 6.  In the graph key.
        self.pe_rows_in_waves = False
        self.K16 = ops.key16_dtype()  # dtype of the key side's 16-bit buffers (csrc/common.h "key16": fp16 since round 4)
        self.load_state(state_dict)

    # ------------------------------------------------------------------------------------------ weights
    def load_state(self, sd):
        d, L = self.dev, self.L
        self._weights_version = getattr(self, '_weights_version', 0) + 1
        g = lambda k: _t(sd[k], d, F32)
        k16 = ops.pack_key16                                                             # fp32 [N,K] -> key16, fragment-major (key-side kernels)
        w = {}
        dec = 'bbox_head.transformer.decoder.'
        for i in range(L):
            p = f'{dec}layers.{i}.'
            w[f'sa_in_b{i}'] = g(p + 'attentions.0.attn.in_proj_bias')
            w[f'sa_out_b{i}'] = g(p + 'attentions.0.attn.out_proj.bias')
            inw, inb = g(p + 'attentions.1.attn.in_proj_weight'), g(p + 'attentions.1.attn.in_proj_bias')
            w[f'ca_q_b{i}'] = inb[:C].contiguous()
            # packed operands of xattn_qmap / xattn_ctxmap: the K / V in_proj ride on the query side (bf16x3)
            w[f'ca_mapA{i}'], w[f'ca_mapB{i}'] = ops.pack_xattn_maps(inw[C:2 * C].contiguous(), inw[2 * C:].contiguous())
            w[f'ca_v_b{i}'] = inb[2 * C:].contiguous()
            w[f'ca_out_b{i}'] = g(p + 'attentions.1.attn.out_proj.bias')
            w[f'ffn_b1{i}'] = g(p + 'ffns.0.layers.0.0.bias')
            w[f'ffn_b2{i}'] = g(p + 'ffns.0.layers.1.bias')
            # bf16x3 + fragment-major copies (row-fused kernels, fused FFN)
            w[f'sa_in_wx{i}'] = ops.pack_x3(g(p + 'attentions.0.attn.in_proj_weight'))
            w[f'sa_out_wx{i}'] = ops.pack_x3(g(p + 'attentions.0.attn.out_proj.weight'))
            w[f'ca_q_wx{i}'] = ops.pack_x3(inw[:C].contiguous())
            w[f'ca_out_wx{i}'] = ops.pack_x3(g(p + 'attentions.1.attn.out_proj.weight'))
            w[f'ffn_w1x{i}'] = ops.pack_x3(g(p + 'ffns.0.layers.0.0.weight'))
            w[f'ffn_w2x{i}'] = ops.pack_x3(g(p + 'ffns.0.layers.1.weight'))
            for n in range(3):
                w[f'ln{n}_w{i}'] = g(p + f'norms.{n}.weight')
                w[f'ln{n}_b{i}'] = g(p + f'norms.{n}.bias')
        w['post_w'], w['post_b'] = g(dec + 'post_norm.weight'), g(dec + 'post_norm.bias')
        w['qe_b0'], w['qe_b2'] = g('bbox_head.query_embedding.0.bias'), g('bbox_head.query_embedding.2.bias')
        w['qe_w0x'], w['qe_w2x'] = ops.pack_x3(g('bbox_head.query_embedding.0.weight')), ops.pack_x3(g('bbox_head.query_embedding.2.weight'))
        st = lambda fmt: torch.stack([g(fmt.format(l)) for l in range(L)]).contiguous()
        for n in ('0', '3'):
            w[f'cls_w{n}'], w[f'cls_b{n}'] = st('bbox_head.cls_branches.{}.' + n + '.weight'), st('bbox_head.cls_branches.{}.' + n + '.bias')
        for n in ('1', '4'):
            w[f'cls_lnw{n}'], w[f'cls_lnb{n}'] = st('bbox_head.cls_branches.{}.' + n + '.weight'), st('bbox_head.cls_branches.{}.' + n + '.bias')
        w['cls_w6'], w['cls_b6'] = st('bbox_head.cls_branches.{}.6.weight'), st('bbox_head.cls_branches.{}.6.bias')
        for n in ('0', '2', '4'):
            w[f'reg_w{n}'], w[f'reg_b{n}'] = st('bbox_head.reg_branches.{}.' + n + '.weight'), st('bbox_head.reg_branches.{}.' + n + '.bias')
        q = 'query_generator.'
        conv = g(q + 'shared_convs.0.conv.weight').permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()       # [256,256,3,3] -> [out][tap][cin]
        w['qg_conv_wp'] = k16(conv)                                                   # key16, fragment-major: the fused conv kernel
        w['qg_conv_b'] = g(q + 'shared_convs.0.conv.bias')
        w['qg_fc_b'] = g(q + 'shared_fcs.0.bias')
        e0 = g(q + 'extra_enc.0.weight')                                              # [512,1040] -> K padded to 1056
        e0p = torch.zeros((e0.shape[0], 1056), device=d, dtype=F32)
        e0p[:, :e0.shape[1]] = e0
        w['qg_e0_b'], w['qg_e2_b'] = g(q + 'extra_enc.0.bias'), g(q + 'extra_enc.2.bias')
        w['qg_c_w'], w['qg_c_b'] = g(q + 'fc_center.weight'), g(q + 'fc_center.bias')
        # LDS-tiled bf16x3 linears (mv2d_linear_x3)
        w['qg_fc_wx'], w['qg_e0_wx'], w['qg_e2_wx'] = ops.pack_x3(g(q + 'shared_fcs.0.weight')), ops.pack_x3(e0p), ops.pack_x3(g(q + 'extra_enc.2.weight'))
        pe = 'position_encoding.'
        c1 = lambda k: g(pe + k).flatten(1).contiguous()
        pe_names = (('w1a', 'position_encoder.0'), ('w1b', 'position_encoder.2'), ('w2a', 'adapt_pos3d.0'), ('w2b', 'adapt_pos3d.2'),
                    ('wr', 'fpe.conv_reduce'), ('we', 'fpe.conv_expand'))
        for n_, k_ in pe_names:
            w['pe_b' + n_[1:]] = g(pe + k_ + '.bias')
        self._pe_w32 = {n_: c1(k_ + '.weight') for n_, k_ in pe_names}               # fp32 originals (3 MB): _c3 builds split copies on demand
        # fragment-major key16 copies for the fused PE kernel (one allocation, in the order the kernel streams them)
        names = ('wr', 'we', 'w1a', 'w1b')
        packs = [k16(c1(dict(pe_names)[n] + '.weight')).view(-1) for n in names]
        flat, off = torch.cat(packs), 0
        w['pe_pack'] = dict(b1a=w['pe_b1a'], b1b=w['pe_b1b'], br=w['pe_br'], be=w['pe_be'], flat=flat)
        for n, t in zip(names, packs):
            w['pe_pack'][n] = flat[off:off + t.numel()]
            off += t.numel()
        # the sine branch's table is built in fp32-class arithmetic (split-precision linears on the unrounded sine rows; once per (weights, geometry))
        for n_, k_ in pe_names[2:4]:
            w['pe_' + n_ + '_x3'] = ops.pack_x3(c1(k_ + '.weight'))
        if self.exact:
            # bf16 hi / lo fragment-major pairs for the split-precision PE kernel (csrc/pe_x3.hip)
            w['pe_x3'] = dict(b1a=w['pe_b1a'], b1b=w['pe_b1b'], br=w['pe_br'], be=w['pe_be'],
                              **{n_: ops.pack_x3(c1(k_ + '.weight')) for n_, k_ in pe_names if n_ in ('w1a', 'w1b', 'wr', 'we')})
            # first-layer weights with their rows in the order csrc/pe_x3b.hip chains the two layers in registers with
            w['pe_x3']['w1a_p'] = ops.pack_x3_rowperm(c1('position_encoder.0.weight'))
            w['pe_x3']['wr_p'] = ops.pack_x3_rowperm(c1('fpe.conv_reduce.weight'))
            w['qg_conv_wx3'] = ops.pack_key16_x3(conv)
        self.w = w
        for k in ('cls_w0', 'cls_w3', 'reg_w0', 'reg_w2'):                              # [L,256,256] -> bf16x3, fragment-major, stacked over L
            w[k + 'x'] = ops.pack_x3_stack(w[k])
        cls_t = (*w['cls_w0x'], w['cls_b0'], w['cls_lnw1'], w['cls_lnb1'], *w['cls_w3x'], w['cls_b3'], w['cls_lnw4'], w['cls_lnb4'], w['cls_w6'], w['cls_b6'])
        reg_t = (*w['reg_w0x'], w['reg_b0'], *w['reg_w2x'], w['reg_b2'], w['reg_w4'], w['reg_b4'])
        self.cls_ptrs_x3, self.reg_ptrs_x3 = ops.make_ptr_array(list(cls_t)), ops.make_ptr_array(list(reg_t))
        # the same tensors from the last decoder layer on: the launch of the last_stage_heads option (every tensor is stacked over L)
        ll = self.L - 1
        self._last_cls, self._last_reg = [t[ll:] for t in cls_t], [t[ll:] for t in reg_t]
        self.cls_ptrs_x3_last, self.reg_ptrs_x3_last = ops.make_ptr_array(self._last_cls), ops.make_ptr_array(self._last_reg)

    def _c3(self, name):
        """K-concatenated split-precision copy [w_hi | w_hi | w_lo] of a PE weight ('w1a', 'w1b', 'w2a', 'w2b', 'wr', 'we'), built on first use
        on the default route (which only keeps the sine branch's pair; the index-exact route holds all six)."""
        k = 'pe_' + name + '_c3'
        if k not in self.w:
            self.w[k] = ops.cat3_weight(self._pe_w32[name])
        return self.w[k]

    # ------------------------------------------------------------------------------------------ workspace
    def _workspace(self, V, h, w, R, Vg=None):
        """Buffers of one (map shape, R) problem.  The number of RoIs changes from frame to frame in real use, so the STORAGE is
        allocated once per (map shape, R rounded up to a multiple of 32, at least 64) and every exact R only gets a dict of dense views into it
        (kernels index [*, R, *] tensors densely) plus its own hipGraph; staging buffers, calibration cache and the stream-order
        guard are shared by all R of a bucket."""
        Vg = V if Vg is None else Vg                     # views per sample; V = all views of the batch
        key = (V, h, w, R, Vg)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        cap = max(64, -(-R // 32) * 32)
        bkey = (V, h, w, cap, Vg)
        base = self._ws_base.get(bkey)
        if base is None:
            store = []

            def alloc(shape, dt=F32, zero=False, pinned=False):
                shape = (shape,) if isinstance(shape, int) else tuple(shape)
                t = (torch.zeros if zero else torch.empty)(shape, dtype=dt, device='cpu' if pinned else self.dev)
                if pinned:
                    t = t.pin_memory()
                store.append(t)
                return t
            self._build_ws(V, h, w, cap, alloc, Vg)
            base = self._ws_base[bkey] = dict(store=store, shared={})
        it = iter(base['store'])

        def view(shape, dt=F32, zero=False, pinned=False):
            shape = (shape,) if isinstance(shape, int) else tuple(shape)
            n = 1
            for v_ in shape:
                n *= v_
            t = next(it)
            assert t.dtype == dt and t.numel() >= n
            return t.view(-1)[:n].view(shape)
        ws = self._build_ws(V, h, w, R, view, Vg)
        ws['shared'] = base['shared']
        self._ws[key] = ws
        return ws

    def _build_ws(self, V, h, w, R, alloc, Vg):
        d, L = self.dev, self.L
        P = V * h * w
        B = V // Vg                                      # samples sharing every launch
        e = lambda shape, dt=F32: alloc(shape, dt)
        z = lambda shape, dt=F32: alloc(shape, dt, zero=True)
        ws = dict(P=P, B=B, Vg=Vg)
        # calibration blob layout (fp64 tables first, then fp32, then bytes)
        lay, off = {}, 0
        for name, n, dt in [('viewK', V * 16, torch.float64), ('viewE', V * 16, torch.float64), ('img2lidar', V * 16, torch.float64),
                            ('trans', V * Vg * 16, torch.float64), ('coords_w', w, torch.float64), ('coords_h', h, torch.float64),
                            ('coords_d', self.depth_num, torch.float64), ('embeds', 3 * P, F32), ('pad_mask', P, torch.uint8)]:
            sz = n * torch.empty(0, dtype=dt).element_size()
            lay[name] = (off, n, dt)
            off += (sz + 15) // 16 * 16
        ws['blob_layout'], ws['blob_bytes'] = lay, off
        ws['blob_h'] = alloc(off, torch.uint8, pinned=True)
        ws['blob_d'] = e(off, torch.uint8)
        ws['tab'] = {k: ws['blob_d'][o:o + n * torch.empty(0, dtype=dt).element_size()].view(dt) for k, (o, n, dt) in lay.items()}
        # per-frame dynamic inputs: RoI list, per-view offsets, first query row of every sample, per-row time step (T path);
        # one pinned staging buffer -> one H2D copy
        o1, o2, o3 = R * 5, R * 5 + (V + 1), R * 5 + (V + 1) + (B + 1)
        dyn_words = o3 + R
        ws['dyn_h'] = alloc(dyn_words, torch.int32, pinned=True)
        ws['dyn_d'] = e(dyn_words, torch.int32)
        for sfx, buf in (('_h', ws['dyn_h']), ('', ws['dyn_d'])):
            ws['rois' + sfx] = buf[:o1].view(F32).view(R, 5)
            ws['view_start' + sfx] = buf[o1:o2]
            ws['grp_start' + sfx] = buf[o2:o3]
            ws['dt_rows' + sfx] = buf[o3:].view(F32)
        K16 = self.K16
        ws['featcl'] = e((P, C))
        ws['enc'] = z((R, 1056)); ws['minv'] = e((R, 16))
        ws['roi_feat'] = e((R, 49, C), K16)
        ws['enc1'] = e((R, 512)); ws['enc2'] = e((R, C)); ws['center'] = e((R, 3))
        ws['xyz'] = e((R, 3)); ws['ref'] = e((R, 3)); ws['posemb'] = e((R, 384)); ws['qe1'] = e((R, C)); ws['qpos'] = e((R, C))
        ws['match'] = e((R, Vg, self.topk), torch.int32)
        Pp = (P + 15) // 16 * 16
        ws['zbuf'] = z(Pp + 32, torch.uint8)                     # roi_mask | nnz[2] | qt_ctl[2] | grp_ctl[2] | lo8_flag: cleared by ONE fill per frame
        ws['roi_mask'] = ws['zbuf'][:P]
        ws['nnz'] = ws['zbuf'][Pp:Pp + 8].view(torch.int32)
        ws['qt_ctl'] = ws['zbuf'][Pp + 8:Pp + 16].view(torch.int32)      # query-order flags (zeroed with zbuf)
        ws['grp_ctl'] = ws['zbuf'][Pp + 16:Pp + 24].view(torch.int32)    # shared-tile cross attention: union entries allocated | capacity flag
        ws['lo8_flag'] = ws['zbuf'][Pp + 24:Pp + 28].view(torch.int32)   # e4m3 lo rows: a remainder left the format's range in this frame (csrc/common.h lo8_pack4_flag)
        ws['zero_mask'] = z(P, torch.uint8)
        ws['rect'] = e((R, 5), torch.int32); ws['pos2s'] = e(P, torch.int32); ws['s2pos'] = e(P, torch.int32)
        ws['S_dev'] = z(1, torch.int32)
        ws['row_ptr'] = e(R + 1, torch.int32)
        if self.kind == 'T':
            ws['bits'] = e(max(ops.csr_workspace_bytes(R, Vg, h, w) // 4, 1), torch.int32)
            ws['row_count'] = e(R, torch.int32)
            ws['col_cap'] = R * self.col_cap_per_query
            ws['S_kv'] = P
        else:
            ws['col_cap'] = R * (1 + Vg * self.topk) * 49
            ws['S_kv'] = R * 49
            ws['roi_sum'] = e((R, 49, C), K16)
        ws['q_order'] = alloc(R, torch.int32, zero=True) if self.q_order else None
        ws['col_idx'] = e(ws['col_cap'], torch.int32)
        # group tables of the shared-tile cross attention (csrc/xattn_group.hip): g_slot | g_cnt | g_ptr | g_len per group, the groups' union key lists
        ng = ops.xattn_group_max(R, B)
        ucap = ws['col_cap'] + 16 * ng
        ws['grp_tab'] = dict(ng=ng, g=alloc((4, ng), torch.int32, zero=True), ucol=alloc(ucap, torch.int32, zero=True),
                             umask=alloc(ucap, torch.uint8, zero=True), ucap=ucap, ctl=ws['grp_ctl'])
        # key16 rows of the PE block's inputs: frustum [.,192], sine [.,384] (training route only: the inference kernel reads the folded
        # table), feature rows [.,256] (the SE gate's input; the value rows of the T path)
        ws['A1'] = e((P, 3 * self.depth_num), K16); ws['A2'] = e((P, 384), K16)
        ws['Xf_b'] = e((P, C), K16)
        if self.exact:
            # index-exact route: the unrounded fp32 frustum rows of the PE block (the feature rows are read from the map), the lo halves of the
            # key / value rows and RoI cells -- all pre-allocated (no per-frame allocation, no host synchronisation: the route is
            # graph-replayable like the default one)
            ws['xa1'] = e((P, 3 * self.depth_num)); ws['xa2'] = e((P, 384))
            lo8 = self._lo8()
            LO = torch.uint8 if lo8 else K16
            if self.kind == 'T':
                ws['xk_lo'] = z((P, C), LO); ws['xv_lo'] = z((P, C), LO)
                ws['roi_lo'] = e((R, 49, C), K16)                 # lo halves of the RoI cells (conv input)
            else:
                ws['xk_lo'] = z((R * 49, C), LO); ws['xv_lo'] = z((R * 49, C), LO)
                # S path: the value rows ARE the RoI cells (key16 lo rows: one array for both; lo8 rows: the split-precision conv, when it runs, reads its own key16 lo cells)
                ws['roi_lo'] = e((R, 49, C), K16) if lo8 else ws['xv_lo'].view(R, 49, C)
        ws['pe'] = e((P, C)); ws['Xk'] = e((P, C), K16)
        ws['pe_pos'] = z((P, C)) if (self.kind == 'S' and self.exact and self.pe_at_positions) else None      # never-listed rows stay 0 (they only ever meet weight 0)
        # cross attention in the raw key space: per-query operand Qt (key16 hi | lo rows of the 8 per-head maps), per-head context sums z
        ws['Qt'] = e((R, 16 * C), K16); ws['zh'] = e((R, 8 * C))
        # unprojected key / value input rows of the cross attention (key + key_pos, key): shared by all layers
        if self.kind == 'T':
            ws['xk_rows'], ws['xv_rows'] = ws['Xk'], ws['Xf_b']
        else:
            ws['xk_rows'], ws['xv_rows'] = ws['roi_sum'].view(R * 49, C), ws['roi_feat'].view(R * 49, C)
        for n in ('x', 'x1', 'x2', 'ctx', 'q'):
            ws[n] = e((R, C))
        ws['zero_rows'] = z((R, C))                              # never written
        ws['sa0_ctx'] = e((R, C))                                # rows of the layer-0 self-attention value bias (fold_sa0), filled per weights version
        ws['qkv'] = e((R, 3 * C)); ws['parts'] = e((2048 // 64, R, C)); ws['outs'] = e((L, R, C))
        ws['cls'] = e((L, R, 10)); ws['reg'] = e((L, R, 10))
        ws['boxes'] = z((B, self.max_num, 9)); ws['scores'] = z((B, self.max_num))
        ws['labels'] = z((B, self.max_num), torch.int64); ws['bbox_index'] = z((B, self.max_num), torch.int64); ws['count'] = z(B, torch.int32)
        return ws

    # ------------------------------------------------------------------------------------------ forward
    def _shape_tables(self, img_metas, h, w):
        """Tables of the padding geometry (frustum grid, padding mask, sine embeds: tens of ms to build), pure functions of the shapes
        they are cached by."""
        skey = (calib.meta_shapes(img_metas), h, w)
        sht = self._shape_cache.get(skey)
        if sht is None:
            sht = calib.shape_tables(skey[0], h, w, stride=self.stride, depth_num=self.depth_num,
                                     position_range=tuple(self.post_range_h64.tolist()))
            if len(self._shape_cache) >= 16:
                self._shape_cache.pop(next(iter(self._shape_cache)))
            self._shape_cache[skey] = sht
        return skey, sht

    def _host_prepare(self, proposals_list, metas_list, V, h, w):
        """Host side of one batch of samples: RoI lists + calibration tables into the workspace's pinned staging buffers.
        The camera matrices are compared with the previous frame's and, when they differ (every frame on the two-frame path: ego motion),
        the derived tables are rebuilt with batched calls and uploaded; the tables of the padding geometry only when the shapes change."""
        B = len(proposals_list)
        Vg = V // B
        # bbox2roi (mmdet) + the dummy proposal rule (RH/mv2d_head.py:105-108), in numpy straight into the staging buffer
        arrs, counts, grp = [], [], [0]
        for b, (props, metas) in enumerate(zip(proposals_list, metas_list)):
            assert len(props) == Vg and len(metas) == Vg, 'every sample of a batch needs the same number of views'
            pa = [(p.detach().to('cpu', F32).numpy() if torch.is_tensor(p) else np.asarray(p, dtype=np.float32)).reshape(-1, 6 if len(p) == 0 else np.shape(p)[-1])
                  for p in props]
            if sum(a.shape[0] for a in pa) == 0:
                pa[0] = np.array([[0, 50, 50, 100, 100, 0]], dtype=np.float32)
            arrs += pa
            counts += [a.shape[0] for a in pa]
            grp.append(grp[-1] + sum(a.shape[0] for a in pa))
        R = grp[-1]
        # The number of RoIs changes with every real frame.  All launches run on the BUCKET size (R rounded up to a multiple of 32, at least 64): the
        # rows R..cap-1 are copies of the last RoI that belong to no sample (not in view_start / grp_start), so nothing attends to them,
        # nothing decodes them and the key set is unchanged; every R of a bucket shares one workspace and ONE captured graph.
        cap = max(64, -(-R // 32) * 32)
        ws = self._workspace(V, h, w, cap, Vg)
        sh = ws['shared']
        if 'done_ev' in sh:
            sh['done_ev'].synchronize()      # the previous frame on this workspace must have consumed the pinned staging buffers (its uploads have run)
        rois_np = ws['rois_h'].numpy()
        rois_np[:R, 0] = np.repeat(np.arange(V, dtype=np.float32), counts)
        rois_np[:R, 1:] = np.concatenate([a[:, :4] for a in arrs if a.shape[0]], 0)
        rois_np[R:] = rois_np[R - 1]
        bh, lay = ws['blob_h'], ws['blob_layout']

        bh_np = bh.numpy()

        def put(k, src):
            # numpy, not Tensor.copy_: above 32768 elements torch spreads a CPU copy over all host threads, and waking 128-256 of them costs
            # milliseconds (measured: 16 two-frame samples per launch -> 3.8 ms per copy, 435 instead of 4500 samples/s on one stream)
            o_, n, dt_ = lay[k]
            nb = n * torch.empty(0, dtype=dt_).element_size()
            a = src.detach().cpu().numpy() if torch.is_tensor(src) else np.asarray(src)
            a = np.ascontiguousarray(a, dtype={torch.float64: np.float64, torch.float32: np.float32, torch.uint8: np.uint8}[dt_])
            bh_np[o_:o_ + nb] = a.reshape(-1).view(np.uint8)
        split = lay['coords_w'][0]                     # [0, split): camera matrices (per frame); [split, end): padding geometry (per rig)
        shp = [self._shape_tables(m, h, w) for m in metas_list]
        shape_key = tuple(k for k, _ in shp)
        fts = [t for _, t in shp]
        f0 = fts[0]
        if sh.get('shape_key') != shape_key:
            assert all(t['pad_h'] == f0['pad_h'] and t['pad_w'] == f0['pad_w'] for t in fts), 'samples of a batch share one pad_shape'
            for k in ('coords_w', 'coords_h', 'coords_d'):
                put(k, f0[k])
            put('embeds', torch.cat([t['embeds'] for t in fts], 1) if B > 1 else f0['embeds'])        # [3, P]: the samples side by side
            put('pad_mask', torch.cat([t['pad_mask'] for t in fts]) if B > 1 else f0['pad_mask'])
            ws['blob_d'][split:].copy_(bh[split:], non_blocking=True)
            sh['shape_key'] = shape_key
        flat = [m for metas in metas_list for m in metas]
        mats = np.stack([np.stack([np.asarray(m[k]) for m in flat]) for k in ('intrinsics', 'extrinsics', 'lidar2img')]).astype(np.float64, copy=False)
        ts = np.array([m.get('timestamp', 0.0) for m in flat], dtype=np.float64).reshape(B, Vg)
        prev = sh.get('mats')
        if prev is None or prev.shape != mats.shape or not np.array_equal(prev, mats):
            # the camera matrices changed: inverse / view-to-view tables rebuilt on the host (batched, ~0.5 ms for 8 x 12 views) and uploaded
            # here, stream-ordered before the frame (23 KB per 12-view sample)
            _, img2lidar, trans, _ = calib.geometry_tables_batch(metas_list)
            put('viewK', torch.from_numpy(mats[0])); put('viewE', torch.from_numpy(mats[1]))
            put('img2lidar', img2lidar); put('trans', trans)
            ws['blob_d'][:split].copy_(bh[:split], non_blocking=True)
            sh['mats'] = mats
        nv = self.num_views
        dts = [float(ts[b, nv:].mean() - ts[b, :nv].mean()) if (self.kind == 'T' and Vg > nv) else 0.0 for b in range(B)]
        sh['frame_scalars'] = dict(pad_h=f0['pad_h'], pad_w=f0['pad_w'], dt=dts[0])
        # the sine branch adapt_pos3d(sine) depends on the padding geometry of the samples only (not on calibration): its table is rebuilt when
        # that (or the weights) change -- in fp32-class arithmetic on both routes (fp32 sine rows, K-concatenated split-precision GEMMs:
        # a once-per-rig cost), so that the table adds no 16-bit rounding of its own to pe
        skey = (self._weights_version,) + tuple(k[0] for k in shape_key)
        if sh.get('sine_key') != skey:
            same = all(k == skey[1] for k in skey[1:])
            P = V * h * w
            Pt = P // B if same else P                                  # one sample's positions when all samples share the geometry
            T, o, W_ = ws['tab'], ops, self.w
            s2 = torch.arange(Pt, dtype=torch.int32, device=self.dev)
            k16e = lambda n_: torch.empty((Pt, n_), device=self.dev, dtype=self.K16)
            a2f = torch.empty((Pt, 384), device=self.dev, dtype=F32)
            a1f = torch.empty((Pt, 3 * self.depth_num), device=self.dev, dtype=F32)
            o.pe_inputs(s2, torch.tensor([Pt], dtype=torch.int32, device=self.dev), Pt, ws['featcl'], T['img2lidar'], T['coords_w'], T['coords_h'],
                        T['coords_d'], T['embeds'], self.const['dim_t'], k16e(3 * self.depth_num), k16e(384), k16e(C), None, V, h, w, self.depth_num,
                        self.post_range_h64, A_frustum_f32=a1f, A_sine_f32=a2f)
            w2a, w2b = self._pe_w32['w2a'], self._pe_w32['w2b']
            h2 = o.linear_x3(a2f, W_['pe_w2a_x3'], W_['pe_b2a'], N=w2a.shape[0], K=w2a.shape[1], act=1)
            tab = o.linear_x3(h2, W_['pe_w2b_x3'], W_['pe_b2b'], N=w2b.shape[0], K=w2b.shape[1])
            del a1f, a2f, h2
            # kept with the tables the workspaces of this map shape share; a captured graph holds the pointer: same shape -> refreshed in place
            if sh.get('sine_tab') is None or sh['sine_tab'].shape != tab.shape:
                sh['sine_tab'] = tab
                sh['sine_gen'] = sh.get('sine_gen', 0) + 1
            else:
                sh['sine_tab'].copy_(tab)
            sh['sine_period'] = Pt
            sh['sine_key'] = skey
        if ws.get('sine_gen') != sh['sine_gen']:
            ws['sine_gen'] = sh['sine_gen']
            ws['graph_stale'] = True                                     # this workspace's graphs were captured with another table
        ws['view_start_h'].numpy()[:] = np.concatenate([[0], np.cumsum(counts)])
        ws['grp_start_h'].numpy()[:] = grp
        if self.kind == 'T':
            # the frame time step is DATA (per row), not a scalar baked into a captured graph: real time stamps differ from frame to frame
            dtr = ws['dt_rows_h'].numpy()
            dtr[:R] = np.repeat(np.asarray(dts, dtype=np.float32), np.diff(grp))
            dtr[R:] = dts[-1]
        sc = dict(sh['frame_scalars'])
        assert max(counts) <= 1024, 'at most 1024 RoIs per view (mv2d_box_correlation)'
        sc['max_per_view'] = max(counts)
        # top-k decode: the kernel sizes its candidate buffer by a power of two >= rows * classes; the launch gets the largest row count
        # of that size class, so that frames with different RoI counts share the launch configuration (and the graph)
        n_pow2 = 1024
        while n_pow2 < max(grp[b + 1] - grp[b] for b in range(B)) * self.num_classes:
            n_pow2 <<= 1
        sc['max_rows'] = min(n_pow2 // self.num_classes, cap)
        sc['cap'] = cap
        return ws, R, sc

    def _tick(self, name):
        if self.prof is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.prof.setdefault(name, []).append(ev)

    def _enqueue(self, ws, feat, R, V, h, w, sc):
        """Device side of one frame: everything below is enqueued on the current stream, no host sync."""
        o, W_ = ops, self.w
        P, L, T = ws['P'], self.L, ws['tab']
        B, Vg = ws['B'], ws['Vg']
        grp = ws['grp_start']                             # first query row of every sample (device): sample-local self attention / top-k; the rows behind the last sample are bucket padding
        tk = self._tick
        tk('h2d')
        rois = ws['rois']                                 # (the RoI list / row tables were uploaded by _run, outside any captured graph)
        # position-major feature map.  Round 5: only the rows somebody reads are transposed (roi_mask: the RoIAlign taps, the key positions and the
        # rows the PE block gates all lie inside some RoI's rectangle), so the transposition runs BEHIND the kernels that build the position list;
        # the whole map is transposed for the training route, for keep_stages runs and when the query-generator chain is forked (key16 mode, T path)
        forked = self.kind == 'T' and self.prof is None and self.fork_qg and not self.exact
        # (T path: roi_mask = the correlation rectangles, which contain every RoIAlign tap only when expand_stride >= 1 -- with the default 0 of
        #  BoxCorrelation a bilinear tap can lie one cell outside its RoI's rectangle; the S path marks the exact tap range.  ADVICE r5)
        masked = (self.masked_transpose and not forked and not self.keep_sine_rows and not getattr(self, '_stage_outputs', False) and
                  (self.kind == 'S' or self.expand >= 1.0) and (h * w) % 4 == 0 and not (torch.is_tensor(feat) and feat.is_contiguous(memory_format=torch.channels_last) and not feat.is_contiguous()))

        def transpose(mask):
            if isinstance(feat, (list, tuple)):
                fcl, Pg = ws['featcl'], P // B
                for b_, f in enumerate(feat):
                    o.nchw_to_nhwc(f, fcl[b_ * Pg:(b_ + 1) * Pg], mask=None if mask is None else mask[b_ * Pg:(b_ + 1) * Pg])
                return fcl
            if feat.is_contiguous(memory_format=torch.channels_last) and not feat.is_contiguous():
                return feat.permute(0, 2, 3, 1).reshape(P, C)                          # already position-major: no copy
            return o.nchw_to_nhwc(feat, ws['featcl'], mask=mask)
        tk('transpose')
        featcl = ws['featcl'] if masked else transpose(None)
        ws['featcl_cur'], ws['map_shape'], ws['max_rows'] = featcl, (V, h, w), sc['max_rows']
        ws['feat_in'], ws['featcl_masked'] = feat, masked            # (train_forward: the masked transposition leaves rows outside every RoI rectangle unwritten)
        tk('box_params'); tk('box_corr')
        # a3/a5/a7 per-RoI camera + a9 epipolar correlation (both independent of the features) + the clearing of the frame's mask / flag
        # bytes: one launch (round 4; a one-sample frame is bound by its NUMBER of kernels)
        o.frame_geometry(rois, T['viewK'], T['viewE'], ws['enc'][:, 1024:], 1056, ws['minv'], ws['view_start'], T['trans'], self.const['lin'],
                         self.const['depths'], ws['match'], Vg, self.topk, sc['pad_h'], sc['pad_w'], sc['max_per_view'], iou_thr=self.iou_thr,
                         ratio=self.ratio, zero=ws['zbuf'])
        tk('csr')
        # T path: the query-generator chain (RoIAlign -> conv -> fcs -> ref points -> query_pos) only needs the feature map and
        # the per-RoI cameras, the key chain (correlation -> key list -> PE -> K/V) only the boxes: run them on two streams
        if forked:
            main = torch.cuda.current_stream()
            side = ws.get('side_stream')
            if side is None:
                side = ws['side_stream'] = torch.cuda.Stream(device=self.dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                o.roi_align(featcl, rois, h, w, out0=ws['roi_feat'], R=R)
                self._enqueue_qg(ws, R)
        if self.kind == 'T':
            # a11/a12: key list + CSR, then a4 RoIAlign of the feature half only
            o.mask_compact(rois, ws['match'], T['pad_mask'], ws['roi_mask'], ws['rect'], ws['pos2s'], ws['s2pos'], ws['S_dev'],
                           ws['bits'], ws['row_count'], ws['row_ptr'], ws['col_idx'], ws['nnz'], R, Vg, h, w, self.topk,
                           self.stride, self.expand, col_cap=ws['col_cap'], n_samples=B)
            if self.q_order and ws.get('q_order') is not None:
                o.xattn_query_order(ws['row_ptr'], ws['col_idx'], grp, R, ws['q_order'], ws['qt_ctl'][1:])
            if self._grouped(ws):
                o.xattn_group_tables(ws['row_ptr'], ws['col_idx'], grp, R, ws['grp_tab'], order=ws.get('q_order') if self.q_order else None)
            if masked:
                transpose(ws['roi_mask'])
            if not forked:
                tk('roi_align')
                o.roi_align(featcl, rois, h, w, out0=ws['roi_feat'],
                            out0_lo=ws.get('roi_lo') if (self.exact and 'conv' not in self.exact_skip) else None, R=R)
        else:
            if self.force_nc is not None:
                # the synthetic rig barely correlates RoIs across views, so the S-path sweep over n_c
                # (RoIs per query) substitutes a synthetic correlation list: own RoI + (n_c - 1) others
                fm = ws.get('forced_match')
                if fm is None or fm.shape != ws['match'].shape:
                    assert self.force_nc - 1 <= Vg * self.topk, 'raise corr_topk for this n_c'
                    fmh = torch.full((R, Vg * self.topk), -1, dtype=torch.int32)
                    ar = torch.arange(R, dtype=torch.int32)
                    for j in range(1, self.force_nc):
                        fmh[:, j - 1] = (ar + 37 * j) % R
                    fm = ws['forced_match'] = fmh.view(R, Vg, self.topk).to(self.dev)
                ws['match'].copy_(fm)
            # positions the RoIAlign taps touch (exact ranges, expand_stride < 0: csrc/geometry.hip roi_tap_range) -> PE only there; CSR over the correlated RoIs' feature rows; launch order of
            # the attention blocks (queries ranked by the smallest RoI they list)
            o.roi_positions_csr(rois, ws['zero_mask'], ws['roi_mask'], ws['rect'], ws['pos2s'], ws['s2pos'], ws['S_dev'], R, V, h, w, ws['match'],
                                ws['row_ptr'], ws['col_idx'], ws['nnz'], Vg, self.topk, stride=self.stride, expand_stride=-1.0, grp_start=grp,
                                order=ws.get('q_order') if self.q_order else None, order_flags=ws['qt_ctl'][1:] if self.q_order else None)
            if self._grouped(ws):
                o.xattn_group_tables(ws['row_ptr'], ws['col_idx'], grp, R, ws['grp_tab'], order=ws.get('q_order') if self.q_order else None)
            if masked:
                transpose(ws['roi_mask'])
        md = ws['S_dev']
        # a2: PE at the listed positions only
        if self.exact and 'pe' not in self.exact_skip:
            tk('pe_inputs'); tk('pe_fused')
            self._exact_pe(ws, featcl, P, V, h, w)
        else:
            tk('pe_inputs')
            o.pe_inputs(ws['s2pos'], md, P, featcl, T['img2lidar'], T['coords_w'], T['coords_h'], T['coords_d'], T['embeds'],
                        self.const['dim_t'], ws['A1'], ws['A2'] if self.keep_sine_rows else None, ws['Xf_b'], None,
                        V, h, w, self.depth_num, self.post_range_h64)
            tk('pe_fused')
            # only what the path reads is written: S: pe (RoIAlign reads it; its keys are RoI-aligned rows), T: Xk (nothing reads pe);
            # a keep_stages run writes both
            dbg = getattr(self, '_stage_outputs', False) or self.exact
            o.pe_fused_tab(ws['A1'], ws['Xf_b'], featcl, md, W_['pe_pack'], ws['shared']['sine_tab'], ws['shared']['sine_period'],
                           ws['pe'] if (self.kind == 'S' or dbg) else None, ws['Xk'] if (self.kind == 'T' or dbg) else None, M=P,
                           row_index=ws['s2pos'])
        if self.kind == 'S':
            tk('roi_align')
            # key rows = RoIAlign(feat) + RoIAlign(pe), value rows = RoIAlign(feat) (index-exact route: both as key16 hi + lo pairs)
            at_pos = self._pe_at_pos(ws)
            pe_map, pe_idx = (ws['pe_pos'], None) if at_pos else (ws['pe'], ws['pos2s'])
            if self.exact and self._lo8():
                o.roi_align(featcl, rois, h, w, map1=pe_map, out0=ws['roi_feat'], out1=ws['roi_sum'], map1_index=pe_idx, out1_is_sum=True,
                            out0_lo=ws.get('roi_lo') if 'conv' not in self.exact_skip else None, out0_lo8=ws['xv_lo'], out1_lo8=ws['xk_lo'], lo8_flag=ws['lo8_flag'], R=R)
            else:
                o.roi_align(featcl, rois, h, w, map1=pe_map, out0=ws['roi_feat'], out1=ws['roi_sum'], map1_index=pe_idx, out1_is_sum=True,
                            out0_lo=ws['xv_lo'] if self.exact else None, out1_lo=ws['xk_lo'] if self.exact else None, R=R)
        if self.ablate_zero_lo and self.exact:
            if self._lo8():
                raise ValueError('ablate_zero_lo works on key16 lo rows: set lo8_rows = False')
            if 'v' in self.ablate_zero_lo and ws.get('xv_lo') is not None:
                ws['xv_lo'].zero_()
            if 'k' in self.ablate_zero_lo and ws.get('xk_lo') is not None:
                ws['xk_lo'].zero_()
            if '8' in self.ablate_zero_lo:                   # the lo halves rounded to 8-bit floats: what 256-byte lo rows would carry.  '8f': OCP e4m3 under the
                for nm in ('xk_lo', 'xv_lo'):                # FIXED scale 2^12 (|lo| <= 2^-12 |x|: the format a kernel can write without a reduction); '8z': e4m3fnuz, '8': e5m2fnuz, per-tensor scale
                    t = ws.get(nm)
                    if t is not None:
                        if '8f' in self.ablate_zero_lo:
                            t.copy_(((t.float() * 4096.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / 4096.0).to(t.dtype))
                            continue
                        f8 = torch.float8_e4m3fnuz if '8z' in self.ablate_zero_lo else torch.float8_e5m2fnuz
                        m = t.float().abs().max().clamp_min(1e-30)
                        q8 = torch.exp2(torch.floor(torch.log2((100.0 if f8 == torch.float8_e4m3fnuz else 16000.0) / m)))
                        t.copy_(((t.float() * q8).to(f8).float() / q8).to(t.dtype))
        if not forked:
            self._enqueue_qg(ws, R)
        if forked:
            torch.cuda.current_stream().wait_stream(side)
        if self.stop_before_decoder:
            tk('end')
            return
        # a16-a19: decoder
        tk('decoder')
        self._enqueue_decoder(ws, R)
        tk('heads')
        self._enqueue_heads(ws, R, sc['dt'])
        tk('decode')
        # a21: NMS-free decode of the last layer (one top-k per sample)
        o.decode_topk(ws['cls'][L - 1], ws['reg'][L - 1], R, self.num_classes, self.max_num, self.post_range_h, ws['boxes'], ws['scores'],
                      ws['labels'], ws['bbox_index'], ws['count'], grp_start=grp, max_grp_rows=sc['max_rows'], payload=ws.get('payload_out'))
        tk('end')

    def _exact_pe(self, ws, featcl, P, V, h, w):
        """Index-exact route: the PE block (MU/pe.py:36-48,64-77,150-166) on UNROUNDED fp32 inputs in ONE split-precision launch
        (mv2d_pe_fused_x3, csrc/pe_x3.hip: a_hi w_hi + a_lo w_hi + a_hi w_lo on bf16 MFMAs, hidden layer hi / lo in LDS, sigmoid, gate
        product and sine-table sum in its epilogue; device-side row count S): pe rows into ws['pe'] (S path), key / value rows as key16 hi +
        lo pairs (T path).  Round 3 ran it as four K-concatenated products on the tile GEMM (hidden layer through HBM) + four split passes:
        1.09 ms (S) / 2.17 ms (T) per 16-sample launch.  No host synchronisation, no allocation: graph-replayable."""
        o, W_, T = ops, self.w, ws['tab']
        md = ws['S_dev']
        if self.keep_sine_rows:
            # (training route: it also reads the key16 rows and the sine rows)
            o.pe_inputs(ws['s2pos'], md, P, featcl, T['img2lidar'], T['coords_w'], T['coords_h'], T['coords_d'], T['embeds'],
                        self.const['dim_t'], ws['A1'], ws['A2'], ws['Xf_b'], None, V, h, w, self.depth_num,
                        self.post_range_h64, A_frustum_f32=ws['xa1'], A_sine_f32=ws['xa2'])
        else:
            # the unrounded frustum rows alone (round 5: 156 -> ~50 us per 141 k positions; the feature rows are read from the map by the PE kernel)
            o.pe_frustum_f32(ws['s2pos'], md, P, T['img2lidar'], T['coords_w'], T['coords_h'], T['coords_d'], ws['xa1'], V, h, w, self.depth_num,
                             self.post_range_h64)
        sh = ws['shared']
        rows = self.kind == 'T'
        dbg = getattr(self, '_stage_outputs', False)
        at_pos = self._pe_at_pos(ws)
        (o.pe_fused_x3b if self.pe_rows_in_waves else o.pe_fused_x3)(
            ws['xa1'], featcl, md, W_['pe_x3'], sh['sine_tab'], sh['sine_period'], pe=(ws['pe_pos'] if at_pos else ws['pe']) if (not rows or dbg) else None,
            Xk=(ws['Xk'], ws['xk_lo']) if rows else None, Xv=(ws['Xf_b'], ws['xv_lo']) if rows else None, M=P, row_index=ws['s2pos'], pe_at_index=at_pos, lo8_flag=ws['lo8_flag'] if (rows and self._lo8()) else None)

    def pe_input_rows(self, ws, positions, V, h, w, f32=False):
        """PE input rows (frustum [n,192], sine [n,384]; key16, or unrounded fp32 with f32=True) at the given map positions (int32, device)
        with the calibration tables of the workspace's current frame: the training route needs them for a key position no RoI lists
        (RH/mv2d_t_head.py:80-82)."""
        n = int(positions.numel())
        T, d = ws['tab'], self.dev
        a1 = torch.empty((n, 3 * self.depth_num), device=d, dtype=self.K16)
        a2 = torch.empty((n, 384), device=d, dtype=self.K16)
        xb = torch.empty((n, C), device=d, dtype=self.K16)
        a1f = torch.empty((n, 3 * self.depth_num), device=d, dtype=F32) if f32 else None
        a2f = torch.empty((n, 384), device=d, dtype=F32) if f32 else None
        ops.pe_inputs(positions.contiguous(), torch.tensor([n], dtype=torch.int32, device=d), n, ws['featcl'], T['img2lidar'], T['coords_w'],
                      T['coords_h'], T['coords_d'], T['embeds'], self.const['dim_t'], a1, a2, xb, None, V, h, w, self.depth_num, self.post_range_h64,
                      A_frustum_f32=a1f, A_sine_f32=a2f)
        return (a1f, a2f) if f32 else (a1, a2)

    def _enqueue_qg(self, ws, R):
        """a6-a8, a13: QueryGenerator on the RoI features -> reference points -> query positional embedding."""
        o, W_, tk = ops, self.w, self._tick
        # a6: QueryGenerator: conv3x3 + ReLU + AvgPool2d(7) fused, one block per RoI (index-exact route: in split precision on the hi + lo cells)
        tk('qg_conv_gemm')
        if self.exact and 'conv' not in self.exact_skip:
            o.qg_conv_pool_x3(ws['roi_feat'], ws['roi_lo'], W_['qg_conv_wx3'], W_['qg_conv_b'], ws['x2'], R=R)
        else:
            o.qg_conv_pool(ws['roi_feat'], W_['qg_conv_wp'], W_['qg_conv_b'], ws['x2'], R=R)
        tk('qg_rest')
        o.linear_x3(ws['x2'], W_['qg_fc_wx'], W_['qg_fc_b'], N=1024, K=256, act=1, clamp=5e3, out=ws['enc'], ldc=1056, M=R)
        o.linear_x3(ws['enc'], W_['qg_e0_wx'], W_['qg_e0_b'], N=512, K=1056, act=1, out=ws['enc1'], M=R)
        o.linear_x3(ws['enc1'], W_['qg_e2_wx'], W_['qg_e2_b'], N=256, K=512, act=1, out=ws['enc2'], M=R)
        # fc_center + reference points (a7/a8) + pos2posemb3d + query_embedding (a13) in one row-fused kernel
        o.query_embed_fused_x3(ws['enc2'], W_['qg_c_w'], W_['qg_c_b'], ws['minv'], self.const['dim_t'], self.pc_range_h, W_['qe_w0x'],
                               W_['qe_b0'], W_['qe_w2x'], W_['qe_b2'], ws['center'], ws['xyz'], ws['ref'], ws['posemb'], ws['qpos'], R=R)

    def _enqueue_decoder(self, ws, R):
        """CrossAttentionBoxHead.forward's transformer call on already-prepared inputs (qpos, key / value rows, CSR):
        the "decoder ms/iter" half of the headline metric.  Per layer: self attention (bf16x3 MFMA, K / V through LDS), out_proj + LN +
        cross-attention q projection (row-fused, bf16x3), cross attention in the raw key space (query map, tile kernel, context map), out_proj
        + LN, fused FFN (bf16x3) and its tail (slab sum + LN + post_norm + the next layer's in_proj)."""
        o, W_, L = ops, self.w, self.L
        x = ws['x']
        xk_rows, xv_rows = ws['xk_rows'], ws['xv_rows']
        dbg = ws.get('dbg_logits')
        maps_fused = ((R <= 512) if self.fuse_maps is None else self.fuse_maps) and dbg is None
        xattn_fused = ((self.kind == 'S' and not ws.get('dn')) if self.fuse_xattn is None else self.fuse_xattn) and dbg is None
        xattn_group = self._grouped(ws) and dbg is None
        lo_k = None if 'attn' in self.exact_skip else ws.get('xk_lo')
        lo_v = None if 'attn' in self.exact_skip else ws.get('xv_lo')

        def tile_attn(i):
            if dbg is not None:
                ws['dbg_q'][i].copy_(ws['q'])
            o.xattn_tile(ws['Qt'], xk_rows, xv_rows, ws['row_ptr'], ws['col_idx'], ws['zh'], R, empty_nan=self.empty_nan, waves=self.xattn_waves,
                         Xk_lo=None if 'attn' in self.exact_skip else ws.get('xk_lo'), Xv_lo=None if 'attn' in self.exact_skip else ws.get('xv_lo'),
                         dbg_logits=None if dbg is None else dbg[i],
                         order=ws.get('q_order') if self.q_order else None)

        # the decoder starts from target = 0 (cross_attention_head.py:32): layer 0 reads a constant zero buffer and qpos directly, from
        # layer 1 on x / xq are the buffers the fused FFN tail writes
        x_in, xq_in = ws['zero_rows'], ws['qpos']
        fold0 = self.fold_sa0 and not ws.get('dn') and ws.get('sa0_ctx') is not None
        if fold0 and ws.get('sa0_ver') != self._weights_version:
            # (first run on this workspace after the weights changed: the eager warm-up in front of a graph capture comes through here)
            ws['sa0_ctx'].copy_(W_['sa_in_b0'][2 * C:].expand(ws['sa0_ctx'].shape[0], C))
            ws['sa0_ver'] = self._weights_version
        for i in range(L):
            if i == 1:
                x_in = x
            sa_ctx = ws['ctx']
            if i == 0 and fold0:
                sa_ctx = ws['sa0_ctx']
            else:
                if i == 0:
                    o.linear_x3(xq_in, W_['sa_in_wx0'], W_['sa_in_b0'], N=3 * C, K=C, A2=x_in, n_split=2 * C, out=ws['qkv'], M=R)
                if ws.get('dn'):
                    o.self_attn_dn(ws['qkv'], ws['dn'][0], ws['dn'][1], out=ws['ctx'])      # training: denoising rows first (train_forward)
                else:
                    o.self_attn(ws['qkv'], ws['ctx'], R, grp_start=ws['grp_start'], max_grp_rows=ws.get('max_rows', 0))
            sa_args = (sa_ctx, x_in, W_[f'sa_out_wx{i}'], W_[f'sa_out_b{i}'], (W_[f'ln0_w{i}'], W_[f'ln0_b{i}']), ws['x1'])
            q_args = dict(qpos=ws['qpos'], Wq_x3=W_[f'ca_q_wx{i}'], bq=W_[f'ca_q_b{i}'], qscale=ops.SCALE_Q, M=R)
            if xattn_group:
                o.attn_out_fused_x3(*sa_args, q_out=ws['q'], **q_args)
                o.xattn_group(ws['q'], W_[f'ca_mapA{i}'], W_[f'ca_mapB{i}'], W_[f'ca_v_b{i}'], xk_rows, xv_rows, ws['row_ptr'], ws['grp_tab'], out=ws['ctx'], R=R,
                              empty_nan=self.empty_nan, Xk_lo=lo_k, Xv_lo=lo_v, order=ws.get('q_order') if self.q_order else None)
                o.attn_out_fused_x3(ws['ctx'], ws['x1'], W_[f'ca_out_wx{i}'], W_[f'ca_out_b{i}'], (W_[f'ln1_w{i}'], W_[f'ln1_b{i}']), ws['x2'], M=R)
            elif xattn_fused:
                o.attn_out_fused_x3(*sa_args, q_out=ws['q'], **q_args)
                o.xattn_fused(ws['q'], W_[f'ca_mapA{i}'], W_[f'ca_mapB{i}'], W_[f'ca_v_b{i}'], xk_rows, xv_rows, ws['row_ptr'], ws['col_idx'], out=ws['ctx'], R=R,
                              empty_nan=self.empty_nan, Xk_lo=lo_k, Xv_lo=lo_v, order=ws.get('q_order') if self.q_order else None)
                o.attn_out_fused_x3(ws['ctx'], ws['x1'], W_[f'ca_out_wx{i}'], W_[f'ca_out_b{i}'], (W_[f'ln1_w{i}'], W_[f'ln1_b{i}']), ws['x2'], M=R)
            elif maps_fused:
                o.attn_out_qmap_x3(*sa_args, WA=W_[f'ca_mapA{i}'], Qt=ws['Qt'], **q_args)
                tile_attn(i)
                o.attn_out_zmap_x3(ws['zh'], W_[f'ca_mapB{i}'], W_[f'ca_v_b{i}'], ws['row_ptr'], ws['x1'], W_[f'ca_out_wx{i}'], W_[f'ca_out_b{i}'],
                                   (W_[f'ln1_w{i}'], W_[f'ln1_b{i}']), ws['x2'], empty_nan=self.empty_nan, M=R)
            else:
                o.attn_out_fused_x3(*sa_args, q_out=ws['q'], **q_args)
                o.xattn_qmap(ws['q'], W_[f'ca_mapA{i}'], ws['Qt'], R=R)
                tile_attn(i)
                o.xattn_ctxmap(ws['zh'], W_[f'ca_mapB{i}'], W_[f'ca_v_b{i}'], ws['row_ptr'], ws['ctx'], R, empty_nan=self.empty_nan)
                o.attn_out_fused_x3(ws['ctx'], ws['x1'], W_[f'ca_out_wx{i}'], W_[f'ca_out_b{i}'], (W_[f'ln1_w{i}'], W_[f'ln1_b{i}']), ws['x2'], M=R)
            # eight hidden slices accumulated per block: 4 slabs to write and re-read instead of 32 (round 3: 8371 vs 8289 samples/s for
            # 8 vs 4 slices, and half the slab traffic).  It fixes the summation order, so it is NOT chosen by the row count: a sample's
            # result must not depend on the batch it is in.
            G = 8
            parts = ws['parts'][:ws['parts'].shape[0] // G]
            o.ffn_fused_x3(ws['x2'], W_[f'ffn_w1x{i}'], W_[f'ffn_b1{i}'], W_[f'ffn_w2x{i}'], parts, R, groups=G)
            nxt = i + 1 < L
            o.ffn_out_fused_x3(parts, W_[f'ffn_b2{i}'], ws['x2'], (W_[f'ln2_w{i}'], W_[f'ln2_b{i}']), (W_['post_w'], W_['post_b']),
                               x, ws['qpos'], None, outs=ws['outs'][i], Win_x3=W_[f'sa_in_wx{i + 1}'] if nxt else None,
                               b_in=W_[f'sa_in_b{i + 1}'] if nxt else None, qkv=ws['qkv'] if nxt else None, M=R)

    def _pe_at_pos(self, ws):
        """S path, index-exact route: are the PE rows written at their map positions (ws['pe_pos']) rather than compacted (ws['pe'])?"""
        return (self.kind == 'S' and self.exact and 'pe' not in self.exact_skip and bool(self.pe_at_positions) and ws.get('pe_pos') is not None and
                not getattr(self, '_stage_outputs', False))

    def _lo8(self):
        """Are the lo halves of the key / value rows e4m3 bytes (csrc/common.h "lo8")?  Not with the shared-tile kernel (its LDS-DMA tiles are key16 rows)."""
        return bool(self.lo8_rows) and self.exact and not self.group_xattn

    def _grouped(self, ws):
        """Does this frame's cross attention run on shared key tiles (csrc/xattn_group.hip)?  Needs the group tables of the workspace (the training
        forward's own decoder workspace has none) and no debug output."""
        on = bool(self.group_xattn)
        return on and ws.get('grp_tab') is not None and not self.debug_attn and not ws.get('dn')

    def _enqueue_heads(self, ws, R, dt):
        # a14: every per-layer cls / reg branch + the reference-point tail in ONE launch (row-block fused, bf16x3)
        dt_rows = ws['dt_rows'] if self.kind == 'T' else None
        if self.last_stage_heads and not getattr(self, '_stage_outputs', False):
            # inference needs the branches of the LAST decoder layer only (the reference evaluates all six and reads [-1],
            # cross_attention_head.py:202-242 / RH/mv2d_head.py:170-194); cls / reg of the other layers are then not written
            ll = self.L - 1
            ops.heads_fused_x3(ws['outs'][ll:], self.cls_ptrs_x3_last, self.reg_ptrs_x3_last, ws['ref'], ws['cls'][ll:], ws['reg'][ll:], R, 1,
                               self.pc_range_h, dt, dt_rows=dt_rows)
        else:
            ops.heads_fused_x3(ws['outs'], self.cls_ptrs_x3, self.reg_ptrs_x3, ws['ref'], ws['cls'], ws['reg'], R, self.L, self.pc_range_h, dt,
                               dt_rows=dt_rows)

    def _result(self, ws, R, keep_stages=False, batch=False):
        sel = (lambda t: t) if batch else (lambda t: t[0])
        out = dict(R=R, ws=ws, cls=ws['cls'][:, :R], reg=ws['reg'][:, :R], boxes=sel(ws['boxes']), scores=sel(ws['scores']), labels=sel(ws['labels']),
                   bbox_index=sel(ws['bbox_index']), count=ws['count'] if batch else ws['count'][:1], grp_start=ws['grp_start_h'].clone())
        if keep_stages:
            # copies of the intermediate buffers restricted to the REAL rows (the launches run on the bucket size, see _host_prepare)
            rows0 = ('rois', 'minv', 'enc', 'roi_feat', 'center', 'xyz', 'ref', 'posemb', 'qpos', 'match')
            st = {}
            for kk in rows0 + ('roi_mask', 'pos2s', 's2pos', 'S_dev', 'col_idx', 'pe', 'Xk', 'Xf_b', 'dbg_logits', 'dbg_q'):
                if ws.get(kk) is not None:
                    st[kk] = (ws[kk][:R] if kk in rows0 else ws[kk]).clone()
            for kk in ('outs', 'cls', 'reg'):
                st[kk] = ws[kk][:, :R].clone()
            st['row_ptr'] = ws['row_ptr'][:R + 1].clone()
            st['nnz'] = torch.stack([ws['row_ptr'][R], ws['nnz'][1]])          # allowed pairs of the real rows | capacity-overflow flag
            out['stages'] = st
        return out

    def run(self, feat, proposals, img_metas, keep_stages=False, use_graph=False, payload=None):
        """One sample.  feat [V,256,h,w] fp32 on the GPU (NCHW, or channels_last memory format); proposals list of [n,6].
        Enqueues the frame on the current stream; use_graph replays a captured hipGraph of the same shape.  payload (optional): fp32
        [1, max_num * 11 + 1] on the GPU -- the decode kernel also writes the wire row of the all-gather of decoded boxes (mv2d_amd.dist)
        there (its address is part of the graph key)."""
        return self._run([feat], [proposals], [img_metas], keep_stages, use_graph, batch=False, payload=payload)

    def run_batch(self, feats, proposals_list, metas_list, keep_stages=False, use_graph=False, payload=None):
        """Several samples through ONE sequence of launches (the reference runs one sample per call): feats = list of [V,256,h,w]
        maps (or one stacked [B*V,256,h,w] tensor), proposals_list / metas_list = one entry per sample.  Outputs: cls / reg
        [L,R_total,10] with the samples' queries concatenated (out['grp_start']), boxes [B,max_num,9], scores, labels, count [B]."""
        return self._run(feats, proposals_list, metas_list, keep_stages, use_graph, batch=True, payload=payload)

    def _run(self, feats, proposals_list, metas_list, keep_stages, use_graph, batch, payload=None):
        B = len(proposals_list)
        if payload is not None:
            assert payload.is_cuda and payload.dtype == F32 and payload.is_contiguous() and tuple(payload.shape) == (B, self.max_num * 11 + 1), \
                'payload: fp32 [samples, max_num * 11 + 1] on the GPU'
        stacked = torch.is_tensor(feats)
        fl = [feats] if stacked else list(feats)
        for f in fl:
            assert f.is_cuda and f.dtype == F32 and f.dim() == 4 and f.shape[1] == C
        if stacked or B == 1:
            feat = fl[0]
            if not (feat.is_contiguous(memory_format=torch.channels_last) and not feat.is_contiguous()):
                feat = feat.contiguous()
            V, _, h, w = feat.shape
            ptrs = (feat.data_ptr(),)
        else:
            assert len(fl) == B
            feat = [f.contiguous() for f in fl]
            Vg, _, h, w = feat[0].shape
            assert all(tuple(f.shape) == tuple(feat[0].shape) for f in feat), 'samples of a batch share one map shape'
            V = Vg * B
            ptrs = tuple(f.data_ptr() for f in feat)
        assert V % B == 0
        ws, R, sc = self._host_prepare(proposals_list, metas_list, V, h, w)
        # upload of the per-frame tables (RoI list, view / sample offsets, time steps): stream-ordered before the frame, outside the captured graph.
        # The "staging consumed" event stays at the END of the frame: recording it right behind this copy would let the host run a frame ahead on every
        # stream; measured in round 3: +1 % in long runs but 7400-7900 instead of 8300 samples/s in short ones -- without the host's wait the four
        # streams drift into phase and their wide kernels collide (LOG.md section 8)
        ws['dyn_d'].copy_(ws['dyn_h'], non_blocking=True)
        ws['payload_out'] = payload
        self._stage_outputs = bool(keep_stages)          # intermediate buffers nothing downstream reads (pe on the T path, Xk on the S path)
        Rc = sc['cap']                     # launches run on the bucket size; R = the real rows
        if self.debug_attn:
            assert not use_graph, 'debug_attn: eager runs only'
            ws['dbg_logits'] = torch.zeros((self.L, 8, ws['col_cap']), device=self.dev, dtype=F32)
            ws['dbg_q'] = torch.zeros((self.L, Rc, C), device=self.dev, dtype=F32)
        else:
            ws.pop('dbg_logits', None); ws.pop('dbg_q', None)
        if not use_graph:
            self._enqueue(ws, feat, Rc, V, h, w, sc)
            self._mark_done(ws)
            return dict(self._result(ws, R, keep_stages, batch), dt=sc['dt'])
        # the graph bakes in the input pointers (the producer's output buffers are static under graph replay) and the
        # frame scalars; anything else changing (RoI boxes, calibration tables, feature values) is data.
        gkey = (ptrs, 0 if payload is None else payload.data_ptr(), sc['pad_h'], sc['pad_w'], sc['max_rows'], self._weights_version, self._stage_outputs, self.last_stage_heads,
                self.xattn_waves, self.fuse_maps, self.fuse_xattn, self.group_xattn, self.lo8_rows, self.pe_at_positions, self.pe_rows_in_waves, self.fold_sa0, self.masked_transpose, self.keep_sine_rows, self.force_nc, self.q_order,
                self.fork_qg, self.exact_skip, self.ablate_zero_lo, self.stop_before_decoder)   # load_state() re-allocates the weights; every route option of __init__ is in the key
        graphs = ws.setdefault('graphs', {})             # one graph per (input buffers, frame scalars): a producer that alternates between
        g = graphs.get(gkey)                             # a few static output buffers replays a few graphs, it does not re-capture
        if ws.pop('graph_stale', False):
            graphs.clear()
            g = None
        if g is None:
            prof, self.prof = self.prof, None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._enqueue(ws, feat, Rc, V, h, w, sc)                   # warm-up outside capture (lazy inits)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: another thread of the process (RCCL's proxy, a data loader) may call into HIP while this one captures
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                self._enqueue(ws, feat, Rc, V, h, w, sc)
            self.prof = prof
            if len(graphs) >= 32:      # (input buffers x payload rows a producer cycles through: bench.py replays 12 per engine)
                graphs.pop(next(iter(graphs)))
            graphs[gkey] = (g, feat)
        else:
            g = g[0]
        g.replay()
        self._mark_done(ws)
        return dict(self._result(ws, R, keep_stages, batch), dt=sc['dt'])

    def train_forward(self, out, dn_ref=None, dn_single=0):
        """Training forward of the decoder + heads on top of a finished ``run`` of ONE sample: the decoder runs again
        over [denoising queries | the sample's queries] with the self-attention mask of ``prepare_for_dn`` evaluated in the kernel and the
        denoising rows attending to every key some RoI can see (RH/mv2d_t_head.py:90-98: ``cross_attn_mask.all(dim=0)``;
        RH/mv2d_s_head.py:158-171).  ``dn_ref`` [pad,3] normalised reference points of the denoising queries (``train.prepare_for_dn``),
        ``dn_single`` rows per group.  Returns (all_cls [L,pad+R,10], all_reg [L,pad+R,10]); the denoising rows' velocities are not divided
        by the frame time step (the reference splits them off before ``_bbox_forward`` does that, RH/mv2d_t_head.py:104-110,132-137).
        Without ``dn_ref`` the inference outputs of all layers are returned (use_denoise=False).  Forward only; synchronises."""
        ws, R = out['ws'], out['R']
        L, d, o, W_ = self.L, self.dev, ops, self.w
        if ws['B'] != 1:
            raise ValueError('train_forward: one sample per run (the reference asserts the same, RH/mv2d_s_head.py:249)')
        row_ptr = ws['row_ptr'][:R + 1]
        nnz = int(row_ptr[R].item())
        if int(ws['nnz'][1].item()) != 0:
            raise RuntimeError('mv2d engine: CSR capacity exceeded (raise col_cap_per_query)')
        col_fb = None
        if self.kind == 'T' and bool((row_ptr[1:] == row_ptr[:-1]).any().item()):
            # the reference un-masks the key at map position (view 0, 0, 0) for a RoI without a visible key in training
            # (RH/mv2d_t_head.py:80-82); if no RoI lists that position its key / value rows are appended behind the S listed ones
            from .train import fallback_key_csr
            s0 = int(ws['pos2s'][0].item())
            if s0 < 0:
                s0 = int(ws['S_dev'].item())
                V_, h_, w_ = ws['map_shape']
                # the PE row of that one position, fp32-class (K-concatenated products like the index-exact route; one row, outside any graph)
                a1f, a2f = self.pe_input_rows(ws, torch.zeros(1, dtype=torch.int32, device=d), V_, h_, w_, f32=True)
                # the feature row of map position 0 comes from the INPUT map: the masked transposition (round 5) writes only the rows inside some RoI's
                # rectangle, and position 0 is by construction outside all of them here
                fin = ws['feat_in']
                fin0 = fin[0] if isinstance(fin, (list, tuple)) else fin
                if fin0.is_contiguous(memory_format=torch.channels_last) and not fin0.is_contiguous():
                    f0 = fin0.permute(0, 2, 3, 1).reshape(-1, C)[:1].contiguous()
                else:
                    f0 = fin0[0, :, 0, 0].reshape(1, C).contiguous()
                def mlp1(x32, n1, n2, **kw):
                    h3 = o.gemm_bf16(o.split3_rows(x32), self._c3(n1), W_['pe_b' + n1[1:]], act=1, split3=True)
                    return o.gemm_bf16(h3, self._c3(n2), W_['pe_b' + n2[1:]], out_dtype=F32, **kw)
                gate = mlp1(f0, 'wr', 'we', act=2)
                pg = mlp1(a1f, 'w1a', 'w1b', mul=gate)
                pe0 = mlp1(a2f, 'w2a', 'w2b', add=pg)
                ws['Xk'][s0:s0 + 1].copy_(o.f32_to_key16(pe0 + f0)); ws['Xf_b'][s0:s0 + 1].copy_(o.f32_to_key16(f0))
            row_ptr, col_fb, _ = fallback_key_csr(row_ptr.clone(), ws['col_idx'][:int(row_ptr[R].item())].clone(), s0)
        no_dn = dn_ref is None or dn_ref.shape[0] == 0
        if no_dn and col_fb is None:
            return ws['cls'][:, :R].clone(), ws['reg'][:, :R].clone()
        if no_dn:
            dn_ref = torch.zeros((0, 3), device=d, dtype=F32)
        pad = int(dn_ref.shape[0])
        T = pad + R
        col = ws['col_idx'][:nnz] if col_fb is None else col_fb
        nnz = int(col.numel())
        keys = torch.unique(col).to(torch.int32)               # every key at least one RoI can see (sorted)
        nk = int(keys.numel())
        e = lambda *shape: torch.empty(shape, device=d, dtype=F32)  # noqa: E731
        posemb = o.posemb3d(dn_ref.to(F32).contiguous(), self.const['dim_t'])
        q1 = o.linear_x3(posemb, W_['qe_w0x'], W_['qe_b0'], N=C, K=384, act=1)
        qdn = o.linear_x3(q1, W_['qe_w2x'], W_['qe_b2'], N=C, K=C)
        tws = dict(B=1, Vg=ws['Vg'], dn=(pad, max(int(dn_single), 1)), grp_start=None, xk_rows=ws['xk_rows'], xv_rows=ws['xv_rows'],
                   Qt=torch.empty((T, 16 * C), device=d, dtype=self.K16), zh=e(T, 8 * C),
                   row_ptr=torch.cat([torch.arange(pad, device=d, dtype=torch.int32) * nk, row_ptr + pad * nk]),
                   col_idx=torch.cat([keys.repeat(pad), col]).contiguous(),
                   ref=torch.cat([dn_ref.to(F32), ws['ref'][:R]]).contiguous(), qpos=torch.cat([qdn, ws['qpos'][:R]]).contiguous(),
                   zero_rows=torch.zeros(T, C, device=d), qkv=e(T, 3 * C), parts=e(2048 // 64, T, C), outs=e(L, T, C), cls=e(L, T, 10),
                   reg=e(L, T, 10), dt_rows=torch.cat([torch.zeros(pad, device=d), torch.full((R,), float(out.get('dt', 0.0)), device=d)]))
        for n in ('x', 'x1', 'x2', 'ctx', 'q'):
            tws[n] = e(T, C)
        self._enqueue_decoder(tws, T)
        self._enqueue_heads(tws, T, float(out.get('dt', 0.0)))
        return tws['cls'], tws['reg']

    @staticmethod
    def _mark_done(ws):
        sh = ws['shared']
        if 'done_ev' not in sh:
            sh['done_ev'] = torch.cuda.Event()
        sh['done_ev'].record()

    def clone_shared(self):
        """A second engine sharing the packed weights / constant tables but with its own workspaces, so that several
        frames can be in flight on different HIP streams (one engine per stream)."""
        other = object.__new__(HeadEngine)
        other.__dict__.update(self.__dict__)
        other._ws = {}
        other._ws_base = {}
        other._shape_cache = self._shape_cache      # pure functions of the padding geometry: shared
        other.prof = None
        return other

    @staticmethod
    def _check_capacity(ws):
        if int(ws['nnz'][1].item()) != 0:
            raise RuntimeError('mv2d engine: CSR capacity exceeded (raise col_cap_per_query)')
        if ws.get('grp_ctl') is not None and int(ws['grp_ctl'][1].item()) != 0:
            raise RuntimeError('mv2d engine: union-list capacity of the shared-tile cross attention exceeded (a CSR row lists a key twice?)')
        if ws.get('lo8_flag') is not None and int(ws['lo8_flag'][0].item()) != 0:
            import warnings
            warnings.warn('mv2d engine: key / value rows beyond +-224 in this frame: their 8-bit lo halves saturated (those elements keep key16 precision); '
                          'set HeadEngine.lo8_rows = False for fp16 lo rows', RuntimeWarning, stacklevel=3)

    def results(self, out):
        """Synchronising accessor: sliced (boxes [K,9], scores [K], labels [K]) like simple_test returns."""
        n = int(out['count'][0].item())
        self._check_capacity(out['ws'])
        return out['boxes'][:n], out['scores'][:n], out['labels'][:n]

    def results_batch(self, out):
        """Synchronising accessor of run_batch: one (boxes [K,9], scores [K], labels [K]) per sample."""
        counts = out['count'].tolist()
        self._check_capacity(out['ws'])
        return [(out['boxes'][b, :n], out['scores'][b, :n], out['labels'][b, :n]) for b, n in enumerate(counts)]
