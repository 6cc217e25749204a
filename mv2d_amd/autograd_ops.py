"""Differentiable dense operators of the head's training route on the hand-written HIP kernels.

The reference trains the head in fp32 under torch autograd (``nn.Linear`` / ``nn.LayerNorm`` / mmcv ``FFN`` inside
MU/petr_transformer.py:195-311 and RH/bbox_heads/cross_attention_head.py:118-142).  Here every matrix product of the forward AND the backward
pass is C = op(A) op(B)^T on ``mv2d_gemm_f32x3`` (csrc/gemm_f32x3.hip: the fp32 operands are read in place in either orientation and split into
bf16 hi / lo while they are staged into LDS, three MFMAs per product, ~1e-5 relative, i.e. fp32-class), a linear layer's whole backward is one
``mv2d_linear_bwd_x3`` call, the layer norms run on ``mv2d_row_ln`` / ``mv2d_layer_norm_bwd``, bias gradients on ``mv2d_colsum``: no rocBLAS /
MIOpen kernel is launched.
PyTorch supplies the autograd graph, the element-wise glue (residual adds, dropout masks, ReLU masks) and the parameter containers.

No CPU path: tensors must live on the GPU.
"""
import torch

from . import _lib, ops
from ._lib import check

BF16, F32 = torch.bfloat16, torch.float32


def _p(t):
    return 0 if t is None else t.data_ptr()


_RAW_STREAM, _GET_DEV = getattr(torch._C, '_cuda_getCurrentRawStream', None), getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    # (torch.cuda.current_stream() costs ~5 us of Python per call -- three calls per launch added up to milliseconds per training step)
    if _RAW_STREAM is not None and _GET_DEV is not None:
        return _RAW_STREAM(_GET_DEV())
    return torch.cuda.current_stream().cuda_stream


_WS = {}


def _workspace(nbytes, device):
    """One scratch buffer per (device, stream): the composite entries below use it strictly in stream order, so consecutive products reuse it."""
    key = (device.index, _stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS[key] = torch.empty(int(nbytes * 1.25) + 4096, device=device, dtype=torch.uint8)
    return buf


def matmul_nt(A, B, bias=None, act=0, trans_a=False, trans_b=False):
    """C = act(op(A) op(B)^T + bias) in fp32-class split precision (bf16 hi / lo, three MFMAs per product); op(X) = X^T when trans_x.  A, B fp32
    2-D; bias [N].  One C call (``mv2d_gemm_f32x3``: fp32 operands read in place in either orientation, split while staged into LDS; split-K for few output
    tiles with a long contraction, slabs summed in fixed order)."""
    A = A if A.stride(-1) == 1 else A.contiguous()
    B = B if B.stride(-1) == 1 else B.contiguous()
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    N, Kb = (B.shape[1], B.shape[0]) if trans_b else B.shape
    assert K == Kb, (A.shape, B.shape, trans_a, trans_b)
    assert A.dtype == F32 and B.dtype == F32 and A.is_cuda and B.is_cuda
    if M == 0 or N == 0 or K == 0:
        return torch.zeros((M, N), device=A.device, dtype=F32)
    lib = _lib.load()
    out = torch.empty((M, N), device=A.device, dtype=F32)
    nb = int(lib.mv2d_gemm_f32x3_ws_bytes(M, N, K))
    ws = _workspace(nb, A.device) if nb else None
    check(lib.mv2d_gemm_f32x3(_p(A), A.stride(0), 1 if trans_a else 0, _p(B), B.stride(0), 1 if trans_b else 0, _p(bias), act, _p(out), N, M, N, K,
                              _p(ws), ws.numel() if ws is not None else 0, _stream()), 'mv2d_gemm_f32x3')
    return out


def colsum(x):
    x = x if x.stride(-1) == 1 else x.contiguous()
    lib = _lib.load()
    out = torch.empty(x.shape[1], device=x.device, dtype=F32)
    nch = int(lib.mv2d_colsum_scratch_rows(x.shape[0]))
    scratch = torch.empty((nch, x.shape[1]), device=x.device, dtype=F32) if nch else None
    check(lib.mv2d_colsum(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), _p(scratch), _stream()), 'mv2d_colsum')
    return out


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b), act 0 = none / 1 = ReLU; x [M,K], W [N,K] (nn.Linear layout), b [N] or None.  Forward and the three backward
    products (dx = g W, dW = g^T x, db = column sums of g; g = dy masked by the ReLU) on the HIP GEMM."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if (x2.dtype == F32 and x2.is_contiguous()) else x2.float().contiguous()
        y = matmul_nt(x2, W.float(), None if b is None else b.float(), act)
        ctx.save_for_backward(x2, W, y if act == 1 else None)
        ctx.meta = (x.shape, b is not None, act)
        return y.reshape(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        shape, has_b, act = ctx.meta
        g = dy.reshape(-1, dy.shape[-1])
        g = g if (g.dtype == F32 and g.is_contiguous()) else g.float().contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2]
        M, N, K = g.shape[0], W.shape[0], W.shape[1]
        dev = g.device
        if M == 0:
            return (torch.zeros(shape, device=dev) if need_x else None, torch.zeros_like(W) if need_w else None,
                    torch.zeros(N, device=dev) if need_b else None, None)
        # one C call: ReLU mask of the gradient, dx = g W, dW = g^T x, db = column sums of g (mv2d_linear_bwd_x3)
        Wf = W if (W.dtype == F32 and W.is_contiguous()) else W.float().contiguous()
        dx, dW, db = _linear_bwd(x2, Wf, y if act == 1 else None, g, need_x, need_w, need_b)
        return (dx.reshape(shape) if need_x else None), (dW.to(W.dtype) if need_w else None), db, None


def linear(x, W, b=None, act=0):
    return LinearFn.apply(x, W, b, act)


def _linear_bwd(x2, W, y, g, need_x, need_w, need_b, dW_out=None, db_out=None):
    """One ``mv2d_linear_bwd_x3`` call; dW / db optionally written into caller-provided (row-slice) buffers."""
    lib = _lib.load()
    M, N, K = g.shape[0], W.shape[0], W.shape[1]
    dev = g.device
    dx = torch.empty((M, K), device=dev, dtype=F32) if need_x else None
    dW = (dW_out if dW_out is not None else torch.empty((N, K), device=dev, dtype=F32)) if need_w else None
    db = (db_out if db_out is not None else torch.empty(N, device=dev, dtype=F32)) if need_b else None
    ws = _workspace(int(lib.mv2d_linear_bwd_x3_ws_bytes(M, N, K)), dev)
    check(lib.mv2d_linear_bwd_x3(_p(x2), _p(W), _p(y), _p(g), _p(dx), _p(dW), _p(db), M, N, K, _p(ws), ws.numel(), _stream()), 'mv2d_linear_bwd_x3')
    return dx, dW, db


def _rows(t):
    t = t.reshape(-1, t.shape[-1])
    return t if (t.dtype == F32 and t.is_contiguous()) else t.float().contiguous()


class InProjFn(torch.autograd.Function):
    """The three input projections of ``nn.MultiheadAttention`` (MU/petr_transformer.py:501-508 -> F.multi_head_attention_forward:
    q = (q_in W_q^T + b_q) / sqrt(d), k = k_in W_k^T + b_k, v = v_in W_v^T + b_v with W = in_proj_weight [3C, C]) as ONE autograd node:
    the weight / bias gradients are written straight into the row blocks of one [3C, C] / [3C] buffer (slicing the parameter in Python
    costs a zero-filled full-size gradient + a copy + an accumulation per slice and step)."""

    @staticmethod
    def forward(ctx, q_in, k_in, v_in, W, b, q_scale):
        Cc = W.shape[1]
        Wf = W if (W.dtype == F32 and W.is_contiguous()) else W.float().contiguous()
        bf = b if b.dtype == F32 else b.float()
        xs = [_rows(q_in), _rows(k_in), _rows(v_in)]
        outs = [matmul_nt(xs[i], Wf[i * Cc:(i + 1) * Cc], bf[i * Cc:(i + 1) * Cc]) for i in range(3)]
        if q_scale != 1.0:
            outs[0].mul_(q_scale)
        ctx.save_for_backward(xs[0], xs[1], xs[2], Wf)
        ctx.meta = (q_in.shape, k_in.shape, v_in.shape, q_scale, W.dtype, b.dtype)
        return tuple(outs)

    @staticmethod
    def backward(ctx, gq, gk, gv):
        x0, x1, x2, Wf = ctx.saved_tensors
        sq, sk, sv, q_scale, wdt, bdt = ctx.meta
        Cc = Wf.shape[1]
        dev = Wf.device
        need_w, need_b = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        dW = torch.empty_like(Wf) if need_w else None
        db = torch.empty(3 * Cc, device=dev, dtype=F32) if need_b else None
        dxs = []
        for i, (x, g, shp) in enumerate(((x0, gq, sq), (x1, gk, sk), (x2, gv, sv))):
            g = _rows(g)
            if i == 0 and q_scale != 1.0:
                g = g * q_scale
            if g.shape[0] == 0:
                if need_w: dW[i * Cc:(i + 1) * Cc].zero_()
                if need_b: db[i * Cc:(i + 1) * Cc].zero_()
                dxs.append(torch.zeros(shp, device=dev) if ctx.needs_input_grad[i] else None)
                continue
            dx, _, _ = _linear_bwd(x, Wf[i * Cc:(i + 1) * Cc], None, g, ctx.needs_input_grad[i], need_w, need_b,
                                   dW[i * Cc:(i + 1) * Cc] if need_w else None, db[i * Cc:(i + 1) * Cc] if need_b else None)
            dxs.append(dx.reshape(shp) if dx is not None else None)
        return dxs[0], dxs[1], dxs[2], (dW.to(wdt) if need_w else None), (db.to(bdt) if need_b else None), None


def in_proj(q_in, k_in, v_in, W, b, q_scale=1.0):
    return InProjFn.apply(q_in, k_in, v_in, W, b, q_scale)


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension of 256 (eps 1e-5): forward ``mv2d_row_ln``, backward ``mv2d_layer_norm_bwd``."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x2 = x.reshape(-1, 256).float().contiguous()
        y = ops.row_ln(x2, ln=(w.float().contiguous(), b.float().contiguous()), eps=eps)
        ctx.save_for_backward(x2, w)
        ctx.meta = (x.shape, eps)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        shape, eps = ctx.meta
        lib = _lib.load()
        M = x2.shape[0]
        g = dy.reshape(-1, 256).float().contiguous()
        dx = torch.empty_like(x2)
        nb = max(int(lib.mv2d_layer_norm_bwd_blocks(M)), 1)
        part = torch.empty((2, nb, 256), device=x2.device, dtype=F32)
        dw = torch.empty(256, device=x2.device, dtype=F32)
        db = torch.empty(256, device=x2.device, dtype=F32)
        check(lib.mv2d_layer_norm_bwd(_p(x2), _p(g), _p(w.float().contiguous()), _p(dx), _p(part[0]), _p(part[1]), _p(dw), _p(db), M, float(eps),
                                      _stream()), 'mv2d_layer_norm_bwd')
        return dx.reshape(shape), dw.to(w.dtype), db.to(w.dtype), None


def layer_norm(x, w, b, eps=1e-5):
    assert x.shape[-1] == 256, 'the HIP layer norm is built for 256 channels (every norm of the head)'
    return LayerNormFn.apply(x, w, b, eps)


class MatmulNTFn(torch.autograd.Function):
    """C = A B^T with both operands differentiable (the dense denoising block of the cross attention: per-head logits and P V)."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.save_for_backward(A, B)
        return matmul_nt(A.float(), B.float())

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dC = dC.float().contiguous()
        dA = matmul_nt(dC, B.float(), trans_b=True) if ctx.needs_input_grad[0] else None        # dC [M,N] . B [N,K]
        dB = matmul_nt(dC, A.float(), trans_a=True, trans_b=True) if ctx.needs_input_grad[1] else None   # dC^T [N,M] . A [M,K]
        return dA, dB


def matmul_nt_ad(A, B):
    return MatmulNTFn.apply(A, B)


DECODER_PARAMS = ('attentions.0.attn.in_proj_weight', 'attentions.0.attn.in_proj_bias', 'attentions.0.attn.out_proj.weight',
                  'attentions.0.attn.out_proj.bias', 'norms.0.weight', 'norms.0.bias',
                  'attentions.1.attn.in_proj_weight', 'attentions.1.attn.in_proj_bias', 'attentions.1.attn.out_proj.weight',
                  'attentions.1.attn.out_proj.bias', 'norms.1.weight', 'norms.1.bias',
                  'ffns.0.layers.0.0.weight', 'ffns.0.layers.0.0.bias', 'ffns.0.layers.1.weight', 'ffns.0.layers.1.bias',
                  'norms.2.weight', 'norms.2.bias')        # the per-layer order of mv2d_train_decoder_fwd's pointer table (include/mv2d_hip.h)


class DecoderFn(torch.autograd.Function):
    """The six decoder layers + the shared post_norm (PETRTransformerDecoder, MU/petr_transformer.py:195-311,563-590) as ONE autograd node:
    forward ``mv2d_train_decoder_fwd``, backward ``mv2d_train_decoder_bwd`` (csrc/train_decoder.hip) -- the launch sequence of either
    direction is issued from C, the parameter-gradient products and the key side of the cross attention on side streams.
    ``meta``: dict(sa=(row_ptr, col), sa_t=csr_transpose(...), ca=..., ca_t=..., drops=(p_sa_attn, p_sa_out, p_ca_attn, p_ca_out, p_ffn_act,
    p_ffn_out), seed=int, L=layers); params: 18 L + 2 fp32 tensors in ``DECODER_PARAMS`` order per layer, then post_norm weight / bias.
    Returns the intermediate outputs [L,T,256]."""

    @staticmethod
    def forward(ctx, qpos, key_in, val_in, meta, *params):
        lib = _lib.load()
        L = meta['L']
        assert len(params) == 18 * L + 2
        for t in (qpos, key_in, val_in) + params:
            assert t.is_cuda and t.dtype == F32 and t.is_contiguous(), 'DecoderFn: fp32 contiguous device tensors'
        T, S, Fw = qpos.shape[0], key_in.shape[0], params[12].shape[0]
        dr = meta['drops']
        keys = meta.get('dn_keys')
        pad = int(meta.get('pad', 0)) if keys is not None else 0
        dims = _lib.TdDims(T, S, L, Fw, meta['sa'][1].numel(), meta['ca'][1].numel(), dr[0], dr[1], dr[2], dr[3], dr[4], dr[5],
                           int(meta['seed']) & 0xffffffff, 1e-5, pad, keys.numel() if pad else 0)
        dev = qpos.device
        act = torch.empty(int(lib.mv2d_train_decoder_act_bytes(_lib.C.byref(dims))), device=dev, dtype=torch.uint8)
        ws = torch.empty(int(lib.mv2d_train_decoder_ws_bytes(_lib.C.byref(dims), 0)), device=dev, dtype=torch.uint8)
        outs = torch.empty((L, T, 256), device=dev, dtype=F32)
        ptrs = (_lib.C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        check(lib.mv2d_train_decoder_fwd(_lib.C.addressof(dims), _lib.C.addressof(ptrs), _p(qpos), _p(key_in), _p(val_in), _p(meta['sa'][0]),
                                         _p(meta['sa'][1]), _p(meta['ca'][0]), _p(meta['ca'][1]), _p(keys if pad else None), _p(outs), _p(act), _p(ws),
                                         _stream()), 'mv2d_train_decoder_fwd')
        ctx.save_for_backward(qpos, key_in, val_in, *params)
        ctx.keep = (dims, ptrs, act, meta)
        return outs

    @staticmethod
    def backward(ctx, d_outs):
        lib = _lib.load()
        qpos, key_in, val_in = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        dims, ptrs, act, meta = ctx.keep
        dev = qpos.device
        d_outs = d_outs if (d_outs.dtype == F32 and d_outs.is_contiguous()) else d_outs.float().contiguous()
        grads, gptr = _flat_grads(params, dev)
        ws = torch.empty(int(lib.mv2d_train_decoder_ws_bytes(_lib.C.byref(dims), 1)), device=dev, dtype=torch.uint8)
        d_qpos, d_key, d_val = torch.empty_like(qpos), torch.empty_like(key_in), torch.empty_like(val_in)
        sa, sa_t, ca, ca_t = meta['sa'], meta['sa_t'], meta['ca'], meta['ca_t']
        if ca_t is None:
            ca_t = ops.csr_transpose(ca[0], ca[1], key_in.shape[0])
        check(lib.mv2d_train_decoder_bwd(_lib.C.addressof(dims), _lib.C.addressof(ptrs), _lib.C.addressof(gptr), _p(qpos), _p(key_in), _p(val_in),
                                         _p(sa[0]), _p(sa[1]), _p(sa_t[0]), _p(sa_t[1]), _p(sa_t[2]), _p(ca[0]), _p(ca[1]), _p(ca_t[0]), _p(ca_t[1]),
                                         _p(ca_t[2]), _p(meta.get('dn_keys') if dims.pad else None), _p(d_outs), _p(act), _p(ws), _p(d_qpos), _p(d_key),
                                         _p(d_val), _stream()), 'mv2d_train_decoder_bwd')
        return (d_qpos, d_key, d_val, None) + tuple(grads)


BRANCH_PARAMS = ('cls_branches.{l}.0.weight', 'cls_branches.{l}.0.bias', 'cls_branches.{l}.1.weight', 'cls_branches.{l}.1.bias',
                 'cls_branches.{l}.3.weight', 'cls_branches.{l}.3.bias', 'cls_branches.{l}.4.weight', 'cls_branches.{l}.4.bias',
                 'cls_branches.{l}.6.weight', 'cls_branches.{l}.6.bias', 'reg_branches.{l}.0.weight', 'reg_branches.{l}.0.bias',
                 'reg_branches.{l}.2.weight', 'reg_branches.{l}.2.bias', 'reg_branches.{l}.4.weight', 'reg_branches.{l}.4.bias')


def _flat_grads(params, dev):
    """one flat fp32 buffer + the views the C entries write the parameter gradients into (64-float aligned pieces)"""
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 63) & ~63
    flat = torch.empty(n, device=dev, dtype=F32)
    return [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)], (_lib.C.c_void_p * len(params))(*[flat.data_ptr() + 4 * o for o in offs])


class HeadsFn(torch.autograd.Function):
    """The class / regression branches of all intermediate outputs (cross_attention_head.py:118-142,200-218) as ONE autograd node:
    ``mv2d_train_heads_fwd`` / ``mv2d_train_heads_bwd`` (csrc/train_decoder.hip).  outs [L,T,256]; params: 16 L fp32 tensors in
    ``BRANCH_PARAMS`` order per layer.  Returns (cls [L,T,NC], raw box code [L,T,10])."""

    @staticmethod
    def forward(ctx, outs, *params):
        lib = _lib.load()
        L, T = outs.shape[:2]
        assert len(params) == 16 * L
        for t in (outs,) + params:
            assert t.is_cuda and t.dtype == F32 and t.is_contiguous(), 'HeadsFn: fp32 contiguous device tensors'
        NC = params[8].shape[0]
        dims = _lib.ThDims(T, L, NC, 1e-5)
        dev = outs.device
        act = torch.empty(int(lib.mv2d_train_heads_act_bytes(_lib.C.byref(dims))), device=dev, dtype=torch.uint8)
        ws = torch.empty(int(lib.mv2d_train_heads_ws_bytes(_lib.C.byref(dims), 0)), device=dev, dtype=torch.uint8)
        cls, reg = torch.empty((L, T, NC), device=dev, dtype=F32), torch.empty((L, T, 10), device=dev, dtype=F32)
        ptrs = (_lib.C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        check(lib.mv2d_train_heads_fwd(_lib.C.addressof(dims), _lib.C.addressof(ptrs), _p(outs), _p(cls), _p(reg), _p(act), _p(ws), _stream()),
              'mv2d_train_heads_fwd')
        ctx.save_for_backward(outs, *params)
        ctx.keep = (dims, ptrs, act)
        return cls, reg

    @staticmethod
    def backward(ctx, d_cls, d_reg):
        lib = _lib.load()
        outs, params = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dims, ptrs, act = ctx.keep
        dev = outs.device
        d_cls = d_cls if (d_cls.dtype == F32 and d_cls.is_contiguous()) else d_cls.float().contiguous()
        d_reg = d_reg if (d_reg.dtype == F32 and d_reg.is_contiguous()) else d_reg.float().contiguous()
        grads, gptr = _flat_grads(params, dev)
        ws = torch.empty(int(lib.mv2d_train_heads_ws_bytes(_lib.C.byref(dims), 1)), device=dev, dtype=torch.uint8)
        d_outs = torch.empty_like(outs)
        check(lib.mv2d_train_heads_bwd(_lib.C.addressof(dims), _lib.C.addressof(ptrs), _lib.C.addressof(gptr), _p(outs), _p(d_cls), _p(d_reg), _p(act),
                                       _p(ws), _p(d_outs), _stream()), 'mv2d_train_heads_bwd')
        return (d_outs,) + tuple(grads)


class BoxCodeFn(torch.autograd.Function):
    """Raw code of all layers [L,T,10] + normalised reference points [T,3] -> boxes [L,T,10] (cross_attention_head.py:216-238,
    RH/mv2d_t_head.py:136-140): ``mv2d_box_code_fwd`` / ``mv2d_box_code_bwd``, one launch each (the torch expression of rounds 3-4 was
    ~12 element-wise launches forward and ~25 backward)."""

    @staticmethod
    def forward(ctx, t, ref, pc_range, pad, dt):
        t, ref = _rows3(t), ref.float().contiguous()
        L, T = t.shape[:2]
        out = torch.empty_like(t)
        rng = (_lib.C.c_float * 6)(*[float(v) for v in pc_range])
        check(_lib.load().mv2d_box_code_fwd(_p(t), _p(ref), _p(out), L, T, int(pad), float(dt), _lib.C.addressof(rng), _stream()), 'mv2d_box_code_fwd')
        ctx.save_for_backward(out, ref)
        ctx.meta = (rng, int(pad), float(dt))
        return out

    @staticmethod
    def backward(ctx, g):
        out, ref = ctx.saved_tensors
        rng, pad, dt = ctx.meta
        L, T = out.shape[:2]
        g = _rows3(g)
        d_t = torch.empty_like(out)
        d_ref = torch.empty_like(ref) if ctx.needs_input_grad[1] else None
        check(_lib.load().mv2d_box_code_bwd(_p(g), _p(out), _p(ref), _p(d_t), _p(d_ref), L, T, pad, dt, _lib.C.addressof(rng), _stream()),
              'mv2d_box_code_bwd')
        return d_t, d_ref, None, None, None


def _rows3(t):
    return t if (t.dtype == F32 and t.is_contiguous()) else t.float().contiguous()


class PosEmbFn(torch.autograd.Function):
    """pos2posemb3d of the normalised reference points (MU/pe.py:20-33 as RH/bbox_heads/cross_attention_head.py:150-156 uses it) with the
    gradient the query generator needs: forward ``mv2d_posemb3d`` (one launch), backward from the saved embedding --
    d p = sum_k (g[2k] emb[2k+1] - g[2k+1] emb[2k]) 2 pi / dim_t[2k] per axis (sin' = cos, cos' = -sin)."""

    @staticmethod
    def forward(ctx, ref, dim_t):
        emb = ops.posemb3d(ref.float().contiguous(), dim_t)
        ctx.save_for_backward(emb, dim_t)
        return emb

    @staticmethod
    def backward(ctx, g):
        emb, dim_t = ctx.saved_tensors
        T = emb.shape[0]
        e, g = emb.view(T, 3, 64, 2), g.reshape(T, 3, 64, 2)
        w = 6.283185307179586 / dim_t[0::2]
        d = ((g[..., 0] * e[..., 1] - g[..., 1] * e[..., 0]) * w).sum(-1)            # [T, 3] in the embedding's axis order (y, x, z)
        return torch.stack((d[:, 1], d[:, 0], d[:, 2]), 1), None


def _bgemm(A, lda, sa, ta, B, ldb, sb, tb, C, ldc, sc, M, N, K, batch):
    lib = _lib.load()
    nb = int(lib.mv2d_gemm_f32x3_batched_ws_bytes(M, N, K, batch))
    ws = _workspace(nb, A.device) if nb else None
    check(lib.mv2d_gemm_f32x3_batched(_p(A), lda, sa, 1 if ta else 0, _p(B), ldb, sb, 1 if tb else 0, _p(C), ldc, sc, M, N, K, batch, 1.0, _p(ws),
                                      ws.numel() if ws is not None else 0, _stream()), 'mv2d_gemm_f32x3_batched')


class DenseHeadsAttnFn(torch.autograd.Function):
    """The denoising rows of the cross attention (RH/mv2d_t_head.py:90-98: they see every key some RoI sees -- a DENSE block): for H heads of
    width d, ctx = dropout(softmax(q_h k_h^T)) v_h with q [n, H d] (pre-scaled), k / v [nk, H d].  The five per-head products of forward and
    backward are ONE batched launch each (``mv2d_gemm_f32x3_batched``: head b = a d-column slice, batch stride d, row stride H d); softmax /
    dropout / the softmax backward are element-wise torch kernels over [H, n, nk].  Rounds 3-4 looped over the heads in Python."""

    @staticmethod
    def forward(ctx, q, k, v, H, p_drop):
        import torch.nn.functional as Fnn
        q, k, v = _rows3(q), _rows3(k), _rows3(v)
        n, Cc = q.shape
        nk, d = k.shape[0], Cc // H
        nkp = (nk + 3) & ~3                        # rows of the [H, n, nk] matrices padded to 16 bytes (vector loads in the products); the pad
        S = torch.empty((H, n, nkp), device=q.device, dtype=F32)      # columns hold -inf logits = zero probabilities
        if nkp > nk:
            S[..., nk:] = float('-inf')
        _bgemm(q, Cc, d, False, k, Cc, d, False, S, nkp, n * nkp, n, nk, d, H)                     # logits_h = q_h k_h^T
        P = torch.softmax(S, -1)
        Pd = Fnn.dropout(P, p_drop, True) if p_drop > 0 else P
        out = torch.empty((n, Cc), device=q.device, dtype=F32)
        _bgemm(Pd, nkp, n * nkp, False, v, Cc, d, True, out, Cc, d, n, d, nk, H)                   # ctx_h = Pd_h v_h  (B = v_h^T read transposed)
        ctx.save_for_backward(q, k, v, P, Pd)
        ctx.meta = (H, float(p_drop))
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, P, Pd = ctx.saved_tensors
        H, p_drop = ctx.meta
        n, Cc = q.shape
        nk, d = k.shape[0], Cc // H
        nkp = P.shape[-1]
        g = _rows3(g)
        dP = torch.empty((H, n, nkp), device=q.device, dtype=F32)
        _bgemm(g, Cc, d, False, v, Cc, d, False, dP, nkp, n * nkp, n, nk, d, H)                    # dPd_h = g_h v_h^T
        # dS = P (dP m - rowsum(P dP m)), m = the dropout mask read back from Pd (a kept probability that is exactly 0 has dS = 0 anyway): one launch, in place
        check(_lib.load().mv2d_softmax_bwd_rows(_p(P), _p(Pd), _p(dP), nkp, H * n, nk, 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0, _stream()),
              'mv2d_softmax_bwd_rows')
        dS = dP
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _bgemm(dS, nkp, n * nkp, False, k, Cc, d, True, dq, Cc, d, n, d, nk, H)                    # dq_h = dS_h k_h
        _bgemm(dS, nkp, n * nkp, True, q, Cc, d, True, dk, Cc, d, nk, d, n, H)                     # dk_h = dS_h^T q_h
        _bgemm(Pd, nkp, n * nkp, True, g, Cc, d, True, dv, Cc, d, nk, d, n, H)                     # dv_h = Pd_h^T g_h
        return dq, dk, dv, None, None


class Im2Col3x3Fn(torch.autograd.Function):
    """x [R,49,256] (7 x 7 cells) -> the unfolded input [R*49, 2304] of a 3 x 3 convolution with padding 1, column order (tap, channel):
    ``mv2d_im2col3x3`` / ``mv2d_col2im3x3`` -- one launch per direction."""

    @staticmethod
    def forward(ctx, x):
        x = _rows3(x)
        R = x.shape[0]
        cols = torch.empty((R * 49, 2304), device=x.device, dtype=F32)
        check(_lib.load().mv2d_im2col3x3(_p(x), _p(cols), R, _stream()), 'mv2d_im2col3x3')
        ctx.R = R
        return cols

    @staticmethod
    def backward(ctx, g):
        g = _rows3(g)
        dx = torch.empty((ctx.R, 49, 256), device=g.device, dtype=F32)
        check(_lib.load().mv2d_col2im3x3(_p(g), _p(dx), ctx.R, _stream()), 'mv2d_col2im3x3')
        return dx


class Center2LidarFn(torch.autograd.Function):
    """(u, v, depth) of every RoI + inverse(K_roi E^T) -> normalised reference point (RH/utils/query_generator.py:333-341,
    RH/mv2d_s_head.py:146-152): ``mv2d_center2lidar_fwd`` / ``_bwd``, one launch each; differentiable in c."""

    @staticmethod
    def forward(ctx, c, minv, pc_range):
        c, minv = _rows3(c), _rows3(minv)
        R = c.shape[0]
        ref = torch.empty((R, 3), device=c.device, dtype=F32)
        rng = (_lib.C.c_float * 6)(*[float(v) for v in pc_range])
        check(_lib.load().mv2d_center2lidar_fwd(_p(c), _p(minv), _p(ref), R, _lib.C.addressof(rng), _stream()), 'mv2d_center2lidar_fwd')
        ctx.save_for_backward(c, minv)
        ctx.rng = rng
        return ref

    @staticmethod
    def backward(ctx, g):
        c, minv = ctx.saved_tensors
        dc = torch.empty_like(c)
        check(_lib.load().mv2d_center2lidar_bwd(_p(_rows3(g)), _p(c), _p(minv), _p(dc), c.shape[0], _lib.C.addressof(ctx.rng), _stream()),
              'mv2d_center2lidar_bwd')
        return dc, None, None


class FlashDenseAttnFn(torch.autograd.Function):
    """The same dense block as ``DenseHeadsAttnFn`` on ``mv2d_dense_attn_fwd`` / ``_bwd`` (csrc/dense_attn.hip): the logits of a 16-query x
    32-key tile live in MFMA accumulators only, nothing of size [heads, n, nk] is written.  q [n,256] pre-scaled, k / v [nk,256]; the dropout mask
    is a counter hash of (seed, head, query, key).  This is what ``mv2d_train_decoder_*`` runs for the denoising rows."""

    @staticmethod
    def forward(ctx, q, k, v, p_drop, seed):
        lib = _lib.load()
        q, k, v = _rows3(q), _rows3(k), _rows3(v)
        n, nk = q.shape[0], k.shape[0]
        out = torch.empty_like(q)
        lse = torch.empty((8, n), device=q.device, dtype=F32)
        ws = torch.empty(int(lib.mv2d_dense_attn_ws_bytes(n, nk, 0)), device=q.device, dtype=torch.uint8)
        check(lib.mv2d_dense_attn_fwd(_p(q), _p(k), _p(v), n, nk, float(p_drop), int(seed) & 0xffffffff, _p(out), _p(lse), _p(ws), _stream()),
              'mv2d_dense_attn_fwd')
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.meta = (float(p_drop), int(seed) & 0xffffffff)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        q, k, v, out, lse = ctx.saved_tensors
        p_drop, seed = ctx.meta
        n, nk = q.shape[0], k.shape[0]
        g = _rows3(g)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(int(lib.mv2d_dense_attn_ws_bytes(n, nk, 1)), device=q.device, dtype=torch.uint8)
        check(lib.mv2d_dense_attn_bwd(_p(q), _p(k), _p(v), _p(out), _p(g), _p(lse), n, nk, p_drop, seed, 1.0, _p(dq), _p(dk), _p(dv), _p(ws), _stream()),
              'mv2d_dense_attn_bwd')
        return dq, dk, dv, None, None
