"""Torch-tensor wrappers over the C-ABI (include/mv2d_hip.h).  PyTorch is plumbing only: device memory,
streams.  Every wrapper enqueues on ``torch.cuda.current_stream()`` and never synchronises.

There is no CPU path here: tensors must live on a HIP device and the extension must be loadable.
"""
import math

import torch

from . import _lib
from ._lib import check

BF16 = torch.bfloat16
_K16 = None


def key16_dtype():
    """torch dtype of the key side's 16-bit format (csrc/common.h "key16"): torch.float16 since round 4 (mv2d_key16_format() == 1); a library
    built with -DMV2D_KEY16_BF16 reports 0 -> torch.bfloat16.  Buffers handed to the key-side kernels must have this dtype."""
    global _K16
    if _K16 is None:
        _K16 = torch.float16 if _lib.load().mv2d_key16_format() == 1 else torch.bfloat16
    return _K16


_RAW_STREAM, _GET_DEV = getattr(torch._C, '_cuda_getCurrentRawStream', None), getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    # (torch.cuda.current_stream() costs ~5 us of Python per call -- three calls per launch added up to milliseconds per training step)
    if _RAW_STREAM is not None and _GET_DEV is not None:
        return _RAW_STREAM(_GET_DEV())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.Mv2dHipError(f'{name}: tensor must be on the GPU (no CPU fallback in the product path)')
    if t.dtype != dtype:
        raise _lib.Mv2dHipError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise _lib.Mv2dHipError(f'{name}: tensor must be contiguous')


def _req16(t, name):
    _req(t, key16_dtype(), name)


def _lo_fmt(*los):
    """lo_fmt of include/mv2d_hip.h for a set of lo row arrays: 1 = e4m3 "lo8" rows (torch.uint8, 256 B per row), 0 = key16 rows (or none)."""
    ts = [t for t in los if t is not None]
    if ts and all(t.dtype == torch.uint8 for t in ts):
        for t in ts:
            _req(t, torch.uint8, 'lo8 rows')
        return 1
    for t in ts:
        _req16(t, 'lo rows')
    return 0


LO8_SCALE = 4096.0            # csrc/common.h "lo8": byte = e4m3(key16_lo * 2^12)


def lo8_decode(b):
    """e4m3 "lo8" rows (uint8) -> the key16 lo rows they stand for (exact: every e4m3 value / 2^12 is an fp16 number or subnormal)."""
    return (b.view(torch.float8_e4m3fn).float() / LO8_SCALE).to(key16_dtype())


def lo8_encode(lo):
    """key16 lo rows -> e4m3 "lo8" rows (uint8): round-to-nearest-even of lo * 2^12 clamped to +-448 -- what the kernels write (csrc/common.h lo8_pack4)."""
    return (lo.float() * LO8_SCALE).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def f32_to_key16(x, out=None, with_lo=False):
    """fp32 -> key16 (the key-side format); with_lo: returns (hi, lo) with x ~ hi + lo."""
    _req(x, torch.float32, 'x')
    hi = torch.empty(x.shape, device=x.device, dtype=key16_dtype()) if out is None else out
    lo = torch.empty(x.shape, device=x.device, dtype=key16_dtype()) if with_lo else None
    _req16(hi, 'out')
    check(_lib.load().mv2d_f32_to_key16(_p(x), _p(hi), _p(lo), x.numel(), _stream()), 'mv2d_f32_to_key16')
    return (hi, lo) if with_lo else hi


def gemm_bf16(A, W, bias=None, *, M=None, A2=None, n_split=0, conv3x3=False, m_dev=None, act=0, mul=None, add=None,
              out=None, out_dtype=BF16, c_blk_stride=0, c_blk_cols=0, out2=None, add2=None, lda=None, ldc=None, split3=False,
              add_index=None, add_period=0, k_splits=1, split_stride=0):
    """C = epi(A @ W.T + bias) with bf16 MFMA.  A [M,K] bf16 (or [R,49,256] when conv3x3), W [N,K] bf16."""
    lib = _lib.load()
    _req(A, BF16, 'A'); _req(W, BF16, 'W'); _req(A2, BF16, 'A2')
    _req(bias, torch.float32, 'bias'); _req(mul, torch.float32, 'mul'); _req(add, torch.float32, 'add'); _req(add2, torch.float32, 'add2')
    N, K = W.shape
    if conv3x3:
        Mrows = A.shape[0] * 49
        lda_ = A.shape[-1]                 # channels per cell: 256, or 768 for [hi | lo | hi] cells (index-exact route)
    else:
        Mrows = A.shape[0]
        lda_ = A.stride(0) if lda is None else lda
    M = Mrows if M is None else M
    if out is None and out2 is None:
        out = torch.empty((M, 3 * N if split3 else N), device=A.device, dtype=out_dtype)
    c_bf16 = 1 if (out is not None and out.dtype == BF16) else 0
    ldc_ = (out.stride(0) if (out is not None and out.dim() == 2 and ldc is None) else (ldc or N))
    if split3:
        ldc_ = out.stride(0)
    rc = lib.mv2d_gemm_bf16_ex(_p(A), _p(A2), n_split, 1 if conv3x3 else 0, _p(W), _p(bias), M, N, K, lda_, _p(m_dev), act,
                               _p(mul), mul.stride(0) if mul is not None else 0, _p(add), add.stride(0) if add is not None else 0,
                               _p(out), c_bf16, ldc_, c_blk_stride, c_blk_cols, _p(out2), _p(add2),
                               out2.stride(0) if out2 is not None else 0, add2.stride(0) if add2 is not None else 0, 1 if split3 else 0,
                               _p(add_index), int(add_period), int(k_splits), int(split_stride), _stream())
    check(rc, 'mv2d_gemm_bf16')
    return out if out is not None else out2


def gemm_f32(A, W, bias=None, *, A2=None, n_split=0, split_k=1, act=0, scale=1.0, clamp=0.0, out=None, out_dtype=torch.float32,
             M=None, lda=None, ldc=None, groups=1, a_gs=0, w_gs=0, b_gs=0, c_gs=0):
    """C = epi((A @ W.T + bias) * scale), exact fp32 MFMA.  A [M,K] fp32, W [N,K] fp32."""
    lib = _lib.load()
    _req(A, torch.float32, 'A'); _req(W, torch.float32, 'W'); _req(A2, torch.float32, 'A2'); _req(bias, torch.float32, 'bias')
    N, K = W.shape[-2], W.shape[-1]
    M = A.shape[-2] if M is None else M
    lda_ = A.stride(-2) if lda is None else lda
    if out is None:
        shape = (split_k, M, N) if split_k > 1 else ((groups, M, N) if groups > 1 else (M, N))
        out = torch.empty(shape, device=A.device, dtype=out_dtype)
    ldc_ = ldc if ldc is not None else (out.stride(-2))
    slice_stride = out.stride(0) if (split_k > 1) else 0
    if groups > 1:
        a_gs = a_gs or (A.stride(0) if A.dim() == 3 else 0)
        w_gs = w_gs or W.stride(0)
        b_gs = b_gs or (bias.stride(0) if bias is not None else 0)
        c_gs = c_gs or out.stride(0)
    rc = lib.mv2d_gemm_f32(_p(A), _p(A2), n_split, _p(W), _p(bias), M, N, K, lda_, W.stride(-2), split_k, act, float(scale),
                           float(clamp), _p(out), 1 if out.dtype == BF16 else 0, ldc_, slice_stride, groups, a_gs, w_gs, b_gs,
                           c_gs, _stream())
    check(rc, 'mv2d_gemm_f32')
    return out


def attn_out_fused(ctx, resid, Wo, bo, ln, x_out, *, qpos=None, Wq=None, bq=None, qscale=1.0, q_out=None, M=None, eps=1e-5):
    """x_out = LN(ctx @ Wo.T + bo + resid); optionally q_out = ((x_out + qpos) @ Wq.T + bq) * qscale."""
    M = ctx.shape[0] if M is None else M
    check(_lib.load().mv2d_attn_out_fused(_p(ctx), _p(resid), _p(Wo), _p(bo), _p(ln[0]), _p(ln[1]), _p(x_out), _p(qpos), _p(Wq), _p(bq),
                                          float(qscale), _p(q_out), M, float(eps), _stream()), 'mv2d_attn_out_fused')
    return x_out


def pack_x3(W):
    """fp32 [N,K] weight -> (hi, lo) pair in the query side's split format (q16: fp16 pairs since round 5), each fragment-major, for the
    split-precision ("x3") kernels."""
    hi, lo = split_q16x2(W.contiguous())
    return pack_wfrag(hi), pack_wfrag(lo)


def rowperm32(N, device=None):
    """Row order of a first-layer weight for csrc/pe_x3b.hip: inside every block of 32 rows, packed row 16 t + 4 fg + e holds original row
    8 fg + 4 t + e -- the accumulator tiles 2 s, 2 s + 1 of a lane (columns 4 fg .. 4 fg + 3 of each) are then the hidden columns
    32 s + 8 fg + {0..3}, {4..7}: the B fragment of k-step s of the next layer."""
    r = torch.arange(N, device=device)
    b, t, m = r // 32, (r % 32) // 16, r % 16
    return 32 * b + 8 * (m // 4) + 4 * t + (m % 4)


def pack_x3_rowperm(W):
    """pack_x3 of the rows of W [N,K] in rowperm32 order (N a multiple of 32)."""
    assert W.shape[0] % 32 == 0
    return pack_x3(W[rowperm32(W.shape[0], W.device)].contiguous())


def pack_key16(W):
    """fp32 [N,K] weight -> key16, fragment-major: the weights of the key-side kernels (PE MLPs, query-generator conv)."""
    return pack_wfrag(f32_to_key16(W.contiguous()))


def pack_key16_x3(W):
    """fp32 [N,K] weight -> (hi, lo) key16 pair, each fragment-major (split-precision conv of the index-exact route)."""
    hi, lo = f32_to_key16(W.contiguous(), with_lo=True)
    return pack_wfrag(hi), pack_wfrag(lo)


def attn_out_fused_x3(ctx, resid, Wo_x3, bo, ln, x_out, *, qpos=None, Wq_x3=None, bq=None, qscale=1.0, q_out=None, M=None, eps=1e-5):
    M = ctx.shape[0] if M is None else M
    wq = Wq_x3 if Wq_x3 is not None else (None, None)
    check(_lib.load().mv2d_attn_out_fused_x3(_p(ctx), _p(resid), _p(Wo_x3[0]), _p(Wo_x3[1]), _p(bo), _p(ln[0]), _p(ln[1]), _p(x_out), _p(qpos),
                                             _p(wq[0]), _p(wq[1]), _p(bq), float(qscale), _p(q_out), M, float(eps), _stream()),
          'mv2d_attn_out_fused_x3')
    return x_out


def attn_out_qmap_x3(ctx, resid, Wo_x3, bo, ln, x_out, *, qpos, Wq_x3, bq, qscale, WA, Qt, M=None, eps=1e-5):
    """attn_out_fused_x3 with the query map of the tile cross attention fused in: writes x_out and Qt (see xattn_qmap), not q."""
    M = ctx.shape[0] if M is None else M
    check(_lib.load().mv2d_attn_out_qmap_x3(_p(ctx), _p(resid), _p(Wo_x3[0]), _p(Wo_x3[1]), _p(bo), _p(ln[0]), _p(ln[1]), _p(x_out), _p(qpos),
                                            _p(Wq_x3[0]), _p(Wq_x3[1]), _p(bq), float(qscale), _p(WA[0]), _p(WA[1]), _p(Qt), M, float(eps), _stream()),
          'mv2d_attn_out_qmap_x3')
    return x_out


def attn_out_zmap_x3(z, WB, bv, row_ptr, resid, Wo_x3, bo, ln, x_out, *, empty_nan=True, M=None, eps=1e-5):
    """xattn_ctxmap + attn_out_fused_x3 (no q stage) in one launch: z [M,8,256] -> x_out = LN(ctx @ Wo.T + bo + resid)."""
    M = z.shape[0] if M is None else M
    check(_lib.load().mv2d_attn_out_zmap_x3(_p(z), _p(WB[0]), _p(WB[1]), _p(bv), _p(row_ptr), 1 if empty_nan else 0, _p(resid), _p(Wo_x3[0]),
                                            _p(Wo_x3[1]), _p(bo), _p(ln[0]), _p(ln[1]), _p(x_out), M, float(eps), _stream()), 'mv2d_attn_out_zmap_x3')
    return x_out


def sa_block_fused_x3(qkv, resid, Wo_x3, bo, ln, x_out, *, qpos=None, Wq_x3=None, bq=None, qscale=1.0, q_out=None, M=None, eps=1e-5):
    """self_attn(qkv) -> out_proj + resid -> LN -> x_out [-> (+qpos) q projection -> q_out]; linears in bf16x3."""
    M = qkv.shape[0] if M is None else M
    wq = Wq_x3 if Wq_x3 is not None else (None, None)
    check(_lib.load().mv2d_sa_block_fused_x3(_p(qkv), _p(resid), _p(Wo_x3[0]), _p(Wo_x3[1]), _p(bo), _p(ln[0]), _p(ln[1]), _p(x_out), _p(qpos),
                                             _p(wq[0]), _p(wq[1]), _p(bq), float(qscale), _p(q_out), M, float(eps), _stream()),
          'mv2d_sa_block_fused_x3')
    return x_out


def query_embed_fused_x3(enc2, Wc, bc, minv, dim_t, pc_range_host, W0_x3, b0, W2_x3, b2, center, xyz, ref, posemb, qpos, R=None):
    R = enc2.shape[0] if R is None else R
    check(_lib.load().mv2d_query_embed_fused_x3(_p(enc2), _p(Wc), _p(bc), _p(minv), _p(dim_t), pc_range_host.data_ptr(), _p(W0_x3[0]),
                                                _p(W0_x3[1]), _p(b0), _p(W2_x3[0]), _p(W2_x3[1]), _p(b2), _p(center), _p(xyz), _p(ref),
                                                _p(posemb), _p(qpos), R, _stream()), 'mv2d_query_embed_fused_x3')
    return qpos


def ffn_out_fused_x3(parts, b2, resid, ln, post, x_out, qpos, xq_out, outs=None, Win_x3=None, b_in=None, qkv=None, M=None, eps=1e-5):
    """y = LN(sum(parts) + b2 + resid) -> x_out, xq_out = y + qpos, outs = post_norm(y); qkv = in_proj(xq, xq, y) (bf16x3)."""
    M = x_out.shape[0] if M is None else M
    w = Win_x3 if Win_x3 is not None else (None, None)
    check(_lib.load().mv2d_ffn_out_fused_x3(_p(parts), parts.shape[0], parts.stride(0), _p(b2), _p(resid), _p(ln[0]), _p(ln[1]),
                                            _p(post[0]) if post else None, _p(post[1]) if post else None, _p(x_out), _p(qpos), _p(xq_out),
                                            _p(outs), _p(w[0]), _p(w[1]), _p(b_in), _p(qkv), M, float(eps), _stream()),
          'mv2d_ffn_out_fused_x3')
    return x_out


def make_ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def pack_wfrag_f32(W):
    """fp32 weight [..., N, K] (rows stacked) -> fragment-major copy of the [prod(...)*N, K] matrix (same number of elements
    when the row count is a multiple of 16)."""
    _req(W, torch.float32, 'W')
    W2 = W.reshape(-1, W.shape[-1]).contiguous()
    N, K = W2.shape
    Wp = torch.empty(((N + 15) // 16) * 16 * K, device=W.device, dtype=torch.float32)
    check(_lib.load().mv2d_pack_wfrag_f32(_p(W2), _p(Wp), N, K, K, _stream()), 'mv2d_pack_wfrag_f32')
    return Wp


def heads_fused(outs, cls_ptrs, reg_ptrs, ref, cls, reg, M, L, pc_range_host, dt=0.0, eps=1e-5, dt_rows=None):
    check(_lib.load().mv2d_heads_fused(_p(outs), cls_ptrs, reg_ptrs, _p(ref), _p(cls), _p(reg), M, L, float(eps),
                                       pc_range_host.data_ptr(), float(dt), _p(dt_rows), _stream()), 'mv2d_heads_fused')


def linear_x3(A, W_x3, bias=None, *, N, K, A2=None, n_split=0, act=0, clamp=0.0, out=None, M=None, lda=None, ldc=None, groups=1,
              a_gs=0, w_gs=0, b_gs=0, c_gs=0, m_dev=None, conv3x3=False, mul=None, add=None):
    """out = act(A @ W.T + bias) in bf16x3; W_x3 = pack_x3(W) of the [N,K] weight (K % 32 == 0, N % 16 == 0).  groups > 1: that many
    linears of the same shape in one launch, group g at A + g*a_gs, W + g*w_gs, bias + g*b_gs, out + g*c_gs (elements)."""
    _req(A, torch.float32, 'A'); _req(A2, torch.float32, 'A2'); _req(bias, torch.float32, 'bias')
    _req(mul, torch.float32, 'mul'); _req(add, torch.float32, 'add'); _req(m_dev, torch.int32, 'm_dev')
    if conv3x3:                                                   # A = [R,49,256] RoI cells, implicit 3x3 convolution (K = 9 * 256)
        M = A.shape[0] * 49 if M is None else M
        lda = 256
    M = A.shape[0] if M is None else M
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    ma = mul if mul is not None else add
    check(_lib.load().mv2d_linear_x3_ex(_p(A), _p(A2), n_split, A.stride(0) if lda is None else lda, _p(W_x3[0]), _p(W_x3[1]), _p(bias), _p(out),
                                        out.stride(0) if ldc is None else ldc, M, N, K, act, float(clamp), groups, a_gs, w_gs, b_gs, c_gs,
                                        _p(m_dev), 1 if conv3x3 else 0, _p(mul), _p(add), ma.stride(0) if ma is not None else 0,
                                        _stream()), 'mv2d_linear_x3')
    return out


def split3_rows(a, b=None, out=None, m_dev=None, M=None):
    """fp32 [M,cols] (+ b) -> bf16 [M, 3 cols] = [hi | lo | hi]: the A operand of a split-precision product through gemm_bf16 (weights
    concatenated as [w_hi | w_hi | w_lo], ``cat3_weight``)."""
    _req(a, torch.float32, 'a'); _req(b, torch.float32, 'b'); _req(out, BF16, 'out'); _req(m_dev, torch.int32, 'm_dev')
    M = a.shape[0] if M is None else M
    cols = a.shape[-1]
    if out is None:
        out = torch.empty((M, 3 * cols), device=a.device, dtype=BF16)
    check(_lib.load().mv2d_split3_rows(_p(a), _p(b), _p(out), M, cols, _p(m_dev), _stream()), 'mv2d_split3_rows')
    return out


def cat3_weight(W, taps=1):
    """fp32 weight [N, taps * K] -> bf16 [N, taps * 3 K]: per tap [w_hi | w_hi | w_lo] (the partner of split3_rows)."""
    hi, lo = split_bf16x2(W.contiguous())
    N = W.shape[0]
    hi, lo = hi.view(N, taps, -1), lo.view(N, taps, -1)
    return torch.cat([hi, hi, lo], 2).reshape(N, -1).contiguous()


def split_rows(a, b=None, hi=None, lo=None, m_dev=None, M=None):
    """(hi, lo) key16 rows of a (+ b): fp32 [M,cols] -> two key16 [M,cols]; rows >= *m_dev (int32, device) are left untouched."""
    _req(a, torch.float32, 'a'); _req(b, torch.float32, 'b'); _req16(hi, 'hi'); _req16(lo, 'lo'); _req(m_dev, torch.int32, 'm_dev')
    M = a.shape[0] if M is None else M
    cols = a.shape[-1]
    if hi is None:
        hi = torch.empty((M, cols), device=a.device, dtype=key16_dtype())
    if lo is None:
        lo = torch.empty((M, cols), device=a.device, dtype=key16_dtype())
    check(_lib.load().mv2d_split_rows_key16(_p(a), _p(b), _p(hi), _p(lo), M, cols, _p(m_dev), _stream()), 'mv2d_split_rows_key16')
    return hi, lo


def heads_fused_x3(outs, cls_ptrs, reg_ptrs, ref, cls, reg, M, L, pc_range_host, dt=0.0, eps=1e-5, dt_rows=None):
    """heads_fused with the 256x256 linears in bf16x3; cls_ptrs / reg_ptrs as documented in include/mv2d_hip.h (pack_x3_stack)."""
    check(_lib.load().mv2d_heads_fused_x3(_p(outs), cls_ptrs, reg_ptrs, _p(ref), _p(cls), _p(reg), M, L, float(eps),
                                          pc_range_host.data_ptr(), float(dt), _p(dt_rows), _stream()), 'mv2d_heads_fused_x3')


def pack_x3_stack(W):
    """fp32 [L,256,256] -> (hi, lo): per-layer pack_x3 copies stacked, [L, 65536] bf16 each."""
    packs = [pack_x3(W[l]) for l in range(W.shape[0])]
    return torch.stack([p[0].view(-1) for p in packs]).contiguous(), torch.stack([p[1].view(-1) for p in packs]).contiguous()


def ffn_pack_weights(W1, W2):
    """nn.Linear weights W1 [hidden,256], W2 [256,hidden] -> fragment-major copies (W1p, W2p) for ffn_fused."""
    _req(W1, torch.float32, 'W1'); _req(W2, torch.float32, 'W2')
    W1p, W2p = torch.empty_like(W1), torch.empty_like(W2)
    check(_lib.load().mv2d_ffn_pack_weights(_p(W1), _p(W2), _p(W1p), _p(W2p), W1.shape[0], _stream()), 'mv2d_ffn_pack_weights')
    return W1p, W2p


def ffn_fused(x, W1, b1, W2, slabs=None, M=None):
    """slabs [hidden/64, M, 256] of partial FFN outputs (sum them + b2 + residual with row_ln); W1, W2 = ffn_pack_weights(...)."""
    _req(x, torch.float32, 'x'); _req(W1, torch.float32, 'W1'); _req(W2, torch.float32, 'W2'); _req(b1, torch.float32, 'b1')
    M = x.shape[0] if M is None else M
    hidden = W1.shape[0]
    if slabs is None:
        slabs = torch.empty((hidden // 64, M, 256), device=x.device, dtype=torch.float32)
    check(_lib.load().mv2d_ffn_fused(_p(x), _p(W1), _p(b1), _p(W2), _p(slabs), M, hidden, _stream()), 'mv2d_ffn_fused')
    return slabs


def ffn_fused_x3(x, W1hl, b1, W2hl, slabs=None, M=None, groups=1):
    """ffn_fused in bf16x3 split precision; W1hl / W2hl = pack_x3(W1) / pack_x3(W2).  groups (1, 2, 4) consecutive hidden slices are
    accumulated per block: hidden/64/groups slabs come out."""
    _req(x, torch.float32, 'x'); _req(b1, torch.float32, 'b1')
    for t in (*W1hl, *W2hl):
        _req(t, q16_dtype(), 'W')
    M = x.shape[0] if M is None else M
    hidden = W1hl[0].numel() // 256                  # packed (fragment-major) copies are flat
    if slabs is None:
        slabs = torch.empty((hidden // 64 // groups, M, 256), device=x.device, dtype=torch.float32)
    check(_lib.load().mv2d_ffn_fused_x3(_p(x), _p(W1hl[0]), _p(W1hl[1]), _p(b1), _p(W2hl[0]), _p(W2hl[1]), _p(slabs), M, hidden, groups,
                                        _stream()), 'mv2d_ffn_fused_x3')
    return slabs


def split_bf16x2(w):
    """fp32 tensor -> (hi, lo) bf16 pair with w ~= hi + lo (|err| ~ 2^-17 |w|)."""
    _req(w, torch.float32, 'w')
    hi = torch.empty(w.shape, device=w.device, dtype=BF16)
    lo = torch.empty(w.shape, device=w.device, dtype=BF16)
    check(_lib.load().mv2d_split_bf16x2(_p(w), _p(hi), _p(lo), w.numel(), _stream()), 'mv2d_split_bf16x2')
    return hi, lo


def q16_dtype():
    """torch dtype of the query side's split format (csrc/common.h "q16"): torch.float16 since round 5 (mv2d_q16_format() == 1)."""
    return torch.float16 if _lib.load().mv2d_q16_format() == 1 else BF16


def split_q16x2(w):
    """fp32 tensor -> (hi, lo) pair with w ~= hi + lo in the library's q16 format (fp16: |err| ~ 2^-23 |w|, bf16: 2^-17 |w|)."""
    _req(w, torch.float32, 'w')
    dt = q16_dtype()
    hi = torch.empty(w.shape, device=w.device, dtype=dt)
    lo = torch.empty(w.shape, device=w.device, dtype=dt)
    check(_lib.load().mv2d_split_q16x2(_p(w), _p(hi), _p(lo), w.numel(), _stream()), 'mv2d_split_q16x2')
    return hi, lo


def row_ln(parts, *, bias=None, residual=None, ln=None, relu=False, out=None, addvec=None, out_plus=None, ln2=None, out2=None,
           M=None, eps=1e-5, rows_per_group=0):
    """y = [relu][LN](sum parts + bias + residual); parts [M,256] or [Z,M,256]."""
    lib = _lib.load()
    _req(parts, torch.float32, 'parts')
    if parts.dim() == 2:
        n_parts, stride = 1, 0
        M = parts.shape[0] if M is None else M
    else:
        n_parts, stride = parts.shape[0], parts.stride(0)
        M = parts.shape[1] if M is None else M
    if out is None and out_plus is None and out2 is None:
        out = torch.empty((M, 256), device=parts.device, dtype=torch.float32)
    lw, lb = ln if ln is not None else (None, None)
    l2w, l2b = ln2 if ln2 is not None else (None, None)
    rc = lib.mv2d_row_ln(_p(parts), n_parts, stride, _p(bias), _p(residual), _p(lw), _p(lb), 1 if relu else 0, _p(out),
                         _p(addvec), _p(out_plus), _p(l2w), _p(l2b), _p(out2), M, float(eps), rows_per_group, _stream())
    check(rc, 'mv2d_row_ln')
    return out


def finalize_reg(reg, ref, L, R, pc_range_host, dt=0.0):
    check(_lib.load().mv2d_finalize_reg(_p(reg), _p(ref), L, R, pc_range_host.data_ptr(), float(dt), _stream()), 'mv2d_finalize_reg')


def pe_fused_tab(A1, Xfb, Xf32, m_dev, wp, sine_tab, tab_period, pe, Xk, M=None, row_index=None, shape=1):
    """The PE block of the key side in one launch: pe = sine_tab[position] + position_encoder(A1) * gate(Xf), Xk = key16(pe + Xf32).
    wp: dict with the fragment-major key16 weights 'w1a','w1b','wr','we' (pack_key16) and fp32 biases 'b1a','b1b','br','be'; sine_tab
    [tab_period,256] fp32 (map position -> adapt_pos3d(sine) + bias).  Xk may be None (S path: only pe is needed; the feature rows are then
    not read), pe may be None when Xk is given (T path).  shape: 1 = 96-row blocks (default), 0 = 64-row blocks (bit-identical)."""
    _req16(A1, 'A1'); _req16(Xfb, 'Xfb'); _req(Xf32, torch.float32, 'Xf32'); _req(sine_tab, torch.float32, 'sine_tab'); _req16(Xk, 'Xk')
    M = A1.shape[0] if M is None else M
    check(_lib.load().mv2d_pe_fused_tab2(_p(A1), _p(Xfb), _p(Xf32), _p(row_index), _p(m_dev), M, _p(wp['w1a']), _p(wp['b1a']), _p(wp['w1b']),
                                         _p(wp['b1b']), _p(wp['wr']), _p(wp['br']), _p(wp['we']), _p(wp['be']), _p(sine_tab), int(tab_period),
                                         _p(pe), _p(Xk), int(shape), _stream()), 'mv2d_pe_fused_tab')
    return pe, Xk


def pe_fused_x3(A1, Xmap, m_dev, wx, sine_tab, tab_period, pe=None, Xk=None, Xv=None, M=None, row_index=None, pe_at_index=False, lo8_flag=None):
    """The PE block in split precision on unrounded inputs (index-exact route, csrc/pe_x3.hip).  A1 [M,192] fp32; Xmap fp32 feature rows
    (indexed by row_index when given); wx: dict 'w1a','w1b','wr','we' = pack_x3(weight) pairs + fp32 biases 'b1a','b1b','br','be'; pe [M,256]
    fp32 and / or Xk = (hi, lo), Xv = (hi, lo) key16 [M,256] pairs (key rows pe + feat, value rows feat).  pe_at_index: pe row m goes to row
    row_index[m] of `pe` (a position-indexed map for roi_align without map1_index)."""
    _req(A1, torch.float32, 'A1'); _req(Xmap, torch.float32, 'Xmap'); _req(sine_tab, torch.float32, 'sine_tab'); _req(pe, torch.float32, 'pe')
    _req(row_index, torch.int32, 'row_index'); _req(m_dev, torch.int32, 'm_dev')
    for pair in (Xk, Xv):
        if pair is not None:
            _req16(pair[0], 'hi')
    lo_fmt = _lo_fmt(Xk[1] if Xk else None, Xv[1] if Xv else None)
    for k in ('w1a', 'w1b', 'wr', 'we'):
        _req(wx[k][0], q16_dtype(), k); _req(wx[k][1], q16_dtype(), k)
    M = A1.shape[0] if M is None else M
    xk, xv = Xk or (None, None), Xv or (None, None)
    check(_lib.load().mv2d_pe_fused_x3(_p(A1), _p(Xmap), _p(row_index), _p(m_dev), M, _p(wx['w1a'][0]), _p(wx['w1a'][1]), _p(wx['b1a']),
                                       _p(wx['w1b'][0]), _p(wx['w1b'][1]), _p(wx['b1b']), _p(wx['wr'][0]), _p(wx['wr'][1]), _p(wx['br']),
                                       _p(wx['we'][0]), _p(wx['we'][1]), _p(wx['be']), _p(sine_tab), int(tab_period), _p(pe), _p(xk[0]), _p(xk[1]),
                                       _p(xv[0]), _p(xv[1]), lo_fmt, 1 if (pe_at_index and row_index is not None) else 0, _p(lo8_flag), _stream()), 'mv2d_pe_fused_x3')
    return pe


def pe_fused_x3b(A1, Xmap, m_dev, wx, sine_tab, tab_period, pe=None, Xk=None, Xv=None, M=None, row_index=None, pe_at_index=False, lo8_flag=None):
    """pe_fused_x3 on the second shape of the kernel (csrc/pe_x3b.hip: a wave owns 16 rows through both layers, the hidden layer stays in registers, the
    weights go through an LDS ring): bitwise the same outputs.  wx additionally holds 'w1a_p', 'wr_p' = pack_x3_rowperm of the two first-layer weights."""
    _req(A1, torch.float32, 'A1'); _req(Xmap, torch.float32, 'Xmap'); _req(sine_tab, torch.float32, 'sine_tab'); _req(pe, torch.float32, 'pe')
    _req(row_index, torch.int32, 'row_index'); _req(m_dev, torch.int32, 'm_dev')
    for pair in (Xk, Xv):
        if pair is not None:
            _req16(pair[0], 'hi')
    lo_fmt = _lo_fmt(Xk[1] if Xk else None, Xv[1] if Xv else None)
    for k in ('w1a_p', 'w1b', 'wr_p', 'we'):
        _req(wx[k][0], q16_dtype(), k); _req(wx[k][1], q16_dtype(), k)
    M = A1.shape[0] if M is None else M
    xk, xv = Xk or (None, None), Xv or (None, None)
    check(_lib.load().mv2d_pe_fused_x3b(_p(A1), _p(Xmap), _p(row_index), _p(m_dev), M, _p(wx['w1a_p'][0]), _p(wx['w1a_p'][1]), _p(wx['b1a']),
                                        _p(wx['w1b'][0]), _p(wx['w1b'][1]), _p(wx['b1b']), _p(wx['wr_p'][0]), _p(wx['wr_p'][1]), _p(wx['br']),
                                        _p(wx['we'][0]), _p(wx['we'][1]), _p(wx['be']), _p(sine_tab), int(tab_period), _p(pe), _p(xk[0]), _p(xk[1]),
                                        _p(xv[0]), _p(xv[1]), lo_fmt, 1 if (pe_at_index and row_index is not None) else 0, _p(lo8_flag), _stream()), 'mv2d_pe_fused_x3b')
    return pe


def pack_wfrag(W):
    """row-major 16-bit weight [N,K] (bf16 or key16) -> fragment-major copy of the same dtype (one MFMA fragment = one contiguous 1 KB)."""
    if W.dtype not in (BF16, torch.float16):
        raise _lib.Mv2dHipError(f'pack_wfrag: expected a 16-bit weight, got {W.dtype}')
    _req(W, W.dtype, 'W')
    N, K = W.shape
    Wp = torch.empty(N * K, device=W.device, dtype=W.dtype)
    check(_lib.load().mv2d_pack_wfrag_bf16(_p(W), _p(Wp), N, K, _stream()), 'mv2d_pack_wfrag_bf16')
    return Wp


def qg_conv_pool(roi_feat, W, bias, out, R=None, ld_out=None):
    """out[r] = avgpool7x7(relu(conv3x3(roi_feat[r]) + bias)); roi_feat [R,49,256] key16, W = pack_key16(conv weight [256,2304])."""
    _req16(roi_feat, 'roi_feat'); _req16(W, 'W'); _req(bias, torch.float32, 'bias'); _req(out, torch.float32, 'out')
    R = roi_feat.shape[0] if R is None else R
    check(_lib.load().mv2d_qg_conv_pool(_p(roi_feat), _p(W), _p(bias), _p(out), out.stride(0) if ld_out is None else ld_out, R, _stream()),
          'mv2d_qg_conv_pool')
    return out


def qg_conv_pool_x3(roi_hi, roi_lo, W_x3, bias, out, R=None, ld_out=None):
    """qg_conv_pool in split precision: RoI cells as key16 hi + lo [R,49,256], W_x3 = pack_key16_x3(conv weight [256,2304])."""
    _req16(roi_hi, 'roi_hi'); _req16(roi_lo, 'roi_lo'); _req16(W_x3[0], 'W_hi'); _req16(W_x3[1], 'W_lo'); _req(bias, torch.float32, 'bias'); _req(out, torch.float32, 'out')
    R = roi_hi.shape[0] if R is None else R
    check(_lib.load().mv2d_qg_conv_pool_x3(_p(roi_hi), _p(roi_lo), _p(W_x3[0]), _p(W_x3[1]), _p(bias), _p(out),
                                           out.stride(0) if ld_out is None else ld_out, R, _stream()), 'mv2d_qg_conv_pool_x3')
    return out


def avgpool49(x, out, ld_out, R):
    check(_lib.load().mv2d_avgpool49(_p(x), _p(out), ld_out, R, _stream()), 'mv2d_avgpool49')
    return out


def f32_to_bf16(x, out=None):
    _req(x, torch.float32, 'x')
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=BF16)
    check(_lib.load().mv2d_f32_to_bf16(_p(x), _p(out), x.numel(), _stream()), 'mv2d_f32_to_bf16')
    return out


def nchw_to_nhwc_bf16(x, out=None):
    """[V,C,h,w] fp32 NCHW -> position-major bf16 [V*h*w, C]."""
    _req(x, torch.float32, 'x')
    V, Cn, h, w = x.shape
    if out is None:
        out = torch.empty((V * h * w, Cn), device=x.device, dtype=BF16)
    check(_lib.load().mv2d_nchw_to_nhwc_bf16(_p(x), _p(out), V, Cn, h * w, _stream()), 'mv2d_nchw_to_nhwc_bf16')
    return out


def map_conv3x3(x_cl, Wp, bias, V, h, w, out=None):
    """x_cl [V*h*w,256] bf16 position-major, Wp = pack_wfrag(conv weight as [256, 9*256] ([out][tap][cin])) -> [V*h*w,256] fp32."""
    _req(x_cl, BF16, 'x_cl'); _req(Wp, BF16, 'Wp'); _req(bias, torch.float32, 'bias')
    if out is None:
        out = torch.empty((V * h * w, 256), device=x_cl.device, dtype=torch.float32)
    check(_lib.load().mv2d_map_conv3x3(_p(x_cl), _p(Wp), _p(bias), _p(out), V, h, w, _stream()), 'mv2d_map_conv3x3')
    return out


def nchw_to_nhwc(x, out=None, mask=None):
    """[V,C,h,w] fp32 -> position-major [V*h*w, C] fp32.  mask (uint8 [V*h*w], device; needs `out`): only the rows whose byte is set are written."""
    _req(x, torch.float32, 'x'); _req(mask, torch.uint8, 'mask')
    V, Cn, h, w = x.shape
    if out is None:
        if mask is not None:
            raise _lib.Mv2dHipError('nchw_to_nhwc: the masked form writes into a caller-owned buffer')
        out = torch.empty((V * h * w, Cn), device=x.device, dtype=torch.float32)
    if mask is not None:
        check(_lib.load().mv2d_nchw_to_nhwc_masked(_p(x), _p(out), _p(mask), V, Cn, h * w, _stream()), 'mv2d_nchw_to_nhwc_masked')
    else:
        check(_lib.load().mv2d_nchw_to_nhwc(_p(x), _p(out), V, Cn, h * w, _stream()), 'mv2d_nchw_to_nhwc')
    return out


def self_attn(qkv, out=None, R=None, grp_start=None, max_grp_rows=0, impl=None):
    """FlattenMHSelfAttention core.  grp_start (int32 [n+1], device): first query row of every sample of a batch; attention stays inside
    a sample.  impl: 'x3' (default: bf16 split precision, K / V through LDS) | 'f32' (exact fp32 MFMA, round 1)."""
    _req(qkv, torch.float32, 'qkv')
    R = qkv.shape[0] if R is None else R
    if out is None:
        out = torch.empty((R, 256), device=qkv.device, dtype=torch.float32)
    n = 0 if grp_start is None else grp_start.numel() - 1
    if (impl or 'x3') == 'x3':
        check(_lib.load().mv2d_self_attn_x3_fwd(_p(qkv), _p(out), R, _p(grp_start), n, int(max_grp_rows), 0, 1, _stream()), 'mv2d_self_attn_x3_fwd')
    else:
        check(_lib.load().mv2d_self_attn_fwd(_p(qkv), _p(out), R, _p(grp_start), n, _stream()), 'mv2d_self_attn_fwd')
    return out


def self_attn_dn(qkv, dn_pad, dn_single, out=None, impl='f32'):
    """Self attention of one training sample: the first dn_pad rows are denoising queries in groups of dn_single (prepare_for_dn's mask).
    impl: 'f32' (default on the training route: exact-fp32 MFMA) | 'x3' (the inference kernel with the mask evaluated in it)."""
    _req(qkv, torch.float32, 'qkv')
    R = qkv.shape[0]
    if out is None:
        out = torch.empty((R, 256), device=qkv.device, dtype=torch.float32)
    if impl == 'x3':
        check(_lib.load().mv2d_self_attn_x3_fwd(_p(qkv), _p(out), R, None, 0, 0, int(dn_pad), max(int(dn_single), 1), _stream()), 'mv2d_self_attn_x3_fwd')
    else:
        check(_lib.load().mv2d_self_attn_dn_fwd(_p(qkv), _p(out), R, int(dn_pad), int(dn_single), _stream()), 'mv2d_self_attn_dn_fwd')
    return out


def dn_queries(gt, gt_labels, rnd, scalar, noise_scale, noise_trans, split, num_classes, pc_range, eps=1e-4):
    """gt [G,9] fp32, gt_labels [G] int32, rnd [G*scalar,3] fp32 in [0,1) -> (ref [G*scalar,3], labels int64, boxes [G*scalar,9])."""
    import ctypes
    G = gt.shape[0]
    n = G * scalar
    ref = torch.empty(n, 3, device=gt.device, dtype=torch.float32)
    labels = torch.empty(n, device=gt.device, dtype=torch.int64)
    boxes = torch.empty(n, 9, device=gt.device, dtype=torch.float32)
    rng = (ctypes.c_float * 6)(*[float(x) for x in pc_range])
    check(_lib.load().mv2d_dn_queries(_p(gt), _p(gt_labels), _p(rnd), G, int(scalar), float(noise_scale), float(noise_trans), float(split),
                                      int(num_classes), ctypes.cast(rng, ctypes.c_void_p), float(eps), _p(ref), _p(labels), _p(boxes),
                                      _stream()), 'mv2d_dn_queries')
    return ref, labels, boxes


def sparse_xattn(q, K, V, row_ptr, col_idx, out=None, R=None, dbg_logits=None, empty_nan=True, p_drop=0.0, seed=0):
    _req(q, torch.float32, 'q'); _req(K, BF16, 'K'); _req(V, BF16, 'V')
    _req(row_ptr, torch.int32, 'row_ptr'); _req(col_idx, torch.int32, 'col_idx')
    R = q.shape[0] if R is None else R
    if out is None:
        out = torch.empty((R, 256), device=q.device, dtype=torch.float32)
    check(_lib.load().mv2d_sparse_xattn_fwd_drop(_p(q), _p(K), _p(V), _p(row_ptr), _p(col_idx), _p(out), _p(dbg_logits),
                                                 dbg_logits.stride(0) if dbg_logits is not None else 0, R, 1 if empty_nan else 0, float(p_drop),
                                                 int(seed) & 0xffffffff, _stream()), 'mv2d_sparse_xattn_fwd')
    return out



def pack_xattn_maps(Wk, Wv):
    """Key / value in_proj weights [256,256] fp32 (rows = output channels) -> the packed operands of the tile cross attention
    (csrc/xattn_tile.hip): (WA_hi, WA_lo) for ``xattn_qmap`` and (WB_hi, WB_lo) for ``xattn_ctxmap``, bf16 hi / lo pairs.
      WA [8 heads][16 tiles][64 lanes = 16 g + m][8 e] = Wk[32 h + 8 g + e][32 (t >> 1) + 8 (m >> 2) + 4 (t & 1) + (m & 3)]
      WB [8 heads][8 steps][2 column tiles][64 lanes = 16 g + n][8 e] = Wv[32 h + 16 nt + n][32 s + 8 g + e]"""
    _req(Wk, torch.float32, 'Wk'); _req(Wv, torch.float32, 'Wv')
    d = Wk.device
    ar = lambda n_: torch.arange(n_, device=d)
    h, t, g, m, e = torch.meshgrid(ar(8), ar(16), ar(4), ar(16), ar(8), indexing='ij')
    wa = Wk[32 * h + 8 * g + e, 32 * (t >> 1) + 8 * (m >> 2) + 4 * (t & 1) + (m & 3)].contiguous()
    h, s, nt, g, n_, e = torch.meshgrid(ar(8), ar(8), ar(2), ar(4), ar(16), ar(8), indexing='ij')
    wb = Wv[32 * h + 16 * nt + n_, 32 * s + 8 * g + e].contiguous()
    return split_q16x2(wa.view(-1)), split_q16x2(wb.view(-1))


def xattn_qmap(q, WA, Qt=None, R=None):
    """q [R,256] fp32 (pre-scaled) -> Qt [R,4096] key16: the fragment-major 16 x 256 operand (hi / lo rows of the 8 per-head maps)."""
    _req(q, torch.float32, 'q'); _req(WA[0], q16_dtype(), 'WA_hi'); _req(WA[1], q16_dtype(), 'WA_lo')
    R = q.shape[0] if R is None else R
    if Qt is None:
        Qt = torch.empty((R, 4096), device=q.device, dtype=key16_dtype())
    _req16(Qt, 'Qt')
    check(_lib.load().mv2d_xattn_qmap(_p(q), _p(WA[0]), _p(WA[1]), _p(Qt), R, _stream()), 'mv2d_xattn_qmap')
    return Qt


def xattn_tile(Qt, Xk, Xv, row_ptr, col_idx, out=None, R=None, dbg_logits=None, empty_nan=True, waves=0, Xk_lo=None, Xv_lo=None, order=None):
    """Tile cross attention on the unprojected key / value rows: Qt from xattn_qmap, Xk / Xv [S,256] key16 -> z [R,8,256] fp32.
    Xk_lo / Xv_lo: optional remainders of the rows (index-exact route: fp32-class key side): key16 [S,256], or e4m3 "lo8" rows (uint8 [S,256])."""
    _req16(Qt, 'Qt'); _req16(Xk, 'Xk'); _req16(Xv, 'Xv')
    lo_fmt = _lo_fmt(Xk_lo, Xv_lo)
    _req(row_ptr, torch.int32, 'row_ptr'); _req(col_idx, torch.int32, 'col_idx'); _req(dbg_logits, torch.float32, 'dbg_logits')
    R = Qt.shape[0] if R is None else R
    if out is None:
        out = torch.empty((R, 8, 256), device=Qt.device, dtype=torch.float32)
    _req(order, torch.int32, 'order')
    check(_lib.load().mv2d_xattn_tile_fwd_ordered(_p(Qt), _p(Xk), _p(Xv), _p(Xk_lo), _p(Xv_lo), _p(row_ptr), _p(col_idx), _p(out), _p(dbg_logits),
                                                  dbg_logits.stride(0) if dbg_logits is not None else 0, R, 1 if empty_nan else 0, int(waves),
                                                  _p(order), lo_fmt, _stream()), 'mv2d_xattn_tile_fwd')
    return out


def xattn_fused(q, WA, WB, bv, Xk, Xv, row_ptr, col_idx, out=None, R=None, empty_nan=True, Xk_lo=None, Xv_lo=None, order=None):
    """ctx [R,256] fp32 = xattn_ctxmap(xattn_tile(xattn_qmap(q))) in ONE launch (csrc/xattn_fused.hip: blocks of 8 queries, Qt / z stay on chip);
    bitwise equal to the three calls with waves=1."""
    _req(q, torch.float32, 'q'); _req(bv, torch.float32, 'bv'); _req(row_ptr, torch.int32, 'row_ptr'); _req(col_idx, torch.int32, 'col_idx')
    for t, n_ in ((WA[0], 'WA_hi'), (WA[1], 'WA_lo'), (WB[0], 'WB_hi'), (WB[1], 'WB_lo')):
        _req(t, q16_dtype(), n_)
    _req16(Xk, 'Xk'); _req16(Xv, 'Xv'); _req(order, torch.int32, 'order')
    lo_fmt = _lo_fmt(Xk_lo, Xv_lo)
    R = q.shape[0] if R is None else R
    if out is None:
        out = torch.empty((R, 256), device=q.device, dtype=torch.float32)
    _req(out, torch.float32, 'out')
    check(_lib.load().mv2d_xattn_fused_fwd(_p(q), _p(WA[0]), _p(WA[1]), _p(WB[0]), _p(WB[1]), _p(bv), _p(Xk), _p(Xv), _p(Xk_lo), _p(Xv_lo), _p(row_ptr),
                                           _p(col_idx), _p(out), R, 1 if empty_nan else 0, _p(order), lo_fmt, _stream()), 'mv2d_xattn_fused_fwd')
    return out


def xattn_group_max(R, n_samples):
    """Upper bound of the number of query groups of mv2d_xattn_group_tables (host arithmetic only)."""
    return (R + 7) // 8 + n_samples + 1


def xattn_group_alloc(R, n_samples, col_cap, device):
    """Buffers of the shared-tile cross attention's group tables for up to R query rows of n_samples samples and col_cap CSR entries:
    dict(ng, g [4, ng] int32 = g_slot | g_cnt | g_ptr | g_len, ucol, umask, ucap, ctl [2] int32 = u_total | overflow flag -- the caller zeroes
    ctl before every xattn_group_tables)."""
    ng = xattn_group_max(R, n_samples)
    ucap = col_cap + 16 * ng
    return dict(ng=ng, g=torch.zeros((4, ng), device=device, dtype=torch.int32), ucol=torch.zeros(ucap, device=device, dtype=torch.int32),
                umask=torch.zeros(ucap, device=device, dtype=torch.uint8), ucap=ucap, ctl=torch.zeros(2, device=device, dtype=torch.int32))


def xattn_group_tables(row_ptr, col_idx, grp_start, R, tab, order=None):
    """Per-group union key lists (csrc/xattn_group.hip) of the CSR for groups of 8 consecutive slots of every sample's `order`; tab =
    xattn_group_alloc(...) with tab['ctl'] zeroed."""
    _req(row_ptr, torch.int32, 'row_ptr'); _req(col_idx, torch.int32, 'col_idx'); _req(grp_start, torch.int32, 'grp_start'); _req(order, torch.int32, 'order')
    g, ctl = tab['g'], tab['ctl']
    check(_lib.load().mv2d_xattn_group_tables(_p(row_ptr), _p(col_idx), _p(order), _p(grp_start), grp_start.numel() - 1, R, tab['ng'], _p(g[0]), _p(g[1]),
                                              _p(g[2]), _p(g[3]), _p(tab['ucol']), _p(tab['umask']), tab['ucap'], _p(ctl[:1]), _p(ctl[1:]), _stream()),
          'mv2d_xattn_group_tables')
    return tab


def xattn_group(q, WA, WB, bv, Xk, Xv, row_ptr, tab, out=None, R=None, empty_nan=True, Xk_lo=None, Xv_lo=None, order=None):
    """ctx [R,256] fp32 = the cross attention of xattn_fused with the key tiles of a group of 8 queries shared (one block per group walks the union of
    the group's key lists once; tab = xattn_group_tables(...) built with the same `order`).  Equal to xattn_fused to fp32 rounding."""
    _req(q, torch.float32, 'q'); _req(bv, torch.float32, 'bv'); _req(row_ptr, torch.int32, 'row_ptr'); _req(order, torch.int32, 'order')
    for t, n_ in ((WA[0], 'WA_hi'), (WA[1], 'WA_lo'), (WB[0], 'WB_hi'), (WB[1], 'WB_lo')):
        _req(t, q16_dtype(), n_)
    _req16(Xk, 'Xk'); _req16(Xv, 'Xv'); _req16(Xk_lo, 'Xk_lo'); _req16(Xv_lo, 'Xv_lo')
    R = q.shape[0] if R is None else R
    if out is None:
        out = torch.empty((R, 256), device=q.device, dtype=torch.float32)
    _req(out, torch.float32, 'out')
    g = tab['g']
    check(_lib.load().mv2d_xattn_group_fwd(_p(q), _p(WA[0]), _p(WA[1]), _p(WB[0]), _p(WB[1]), _p(bv), _p(Xk), _p(Xv), _p(Xk_lo), _p(Xv_lo), _p(row_ptr),
                                           _p(order), _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3]), _p(tab['ucol']), _p(tab['umask']), _p(out), tab['ng'],
                                           1 if empty_nan else 0, _stream()), 'mv2d_xattn_group_fwd')
    return out


def xattn_query_order(row_ptr, col_idx, grp_start, R, perm, flags, stride=0):
    """perm [R] int32 = the query rows of every sample (grp_start [n+1], device) sorted by their smallest key (stride 0: the first entry of a
    CSR row; stride 49: the smallest first cell of the RoIs an S-path row lists); flags int32 [>=1] (zeroed by the caller)."""
    check(_lib.load().mv2d_xattn_query_order(_p(row_ptr), _p(col_idx), _p(grp_start), grp_start.numel() - 1, R, _p(perm), _p(flags), int(stride), _stream()),
          'mv2d_xattn_query_order')
    return perm


def xattn_ctxmap(z, WB, bv, row_ptr, out=None, R=None, empty_nan=True):
    """z [R,8,256] fp32 -> ctx [R,256] = Wv_h z_h + bv; rows without a key (row_ptr) give NaN / 0."""
    _req(z, torch.float32, 'z'); _req(WB[0], q16_dtype(), 'WB_hi'); _req(WB[1], q16_dtype(), 'WB_lo'); _req(bv, torch.float32, 'bv')
    _req(row_ptr, torch.int32, 'row_ptr')
    R = z.shape[0] if R is None else R
    if out is None:
        out = torch.empty((R, 256), device=z.device, dtype=torch.float32)
    check(_lib.load().mv2d_xattn_ctxmap(_p(z), _p(WB[0]), _p(WB[1]), _p(bv), _p(row_ptr), _p(out), R, 1 if empty_nan else 0, _stream()),
          'mv2d_xattn_ctxmap')
    return out


def csr_transpose(row_ptr, col_idx, S):
    """The allowed pairs grouped by key (for the key pass of sparse_xattn_bwd): key_ptr [S+1], pair_idx [nnz] (pair ids in CSR order,
    stable within a key), pair_row [nnz] (query of every pair); torch ops on the device of the CSR."""
    nnz = col_idx.numel()
    R = row_ptr.numel() - 1
    counts = (row_ptr[1:] - row_ptr[:-1]).long()
    pair_row = torch.repeat_interleave(torch.arange(R, device=col_idx.device, dtype=torch.int32), counts, output_size=nnz)
    pair_idx = torch.sort(col_idx.long(), stable=True).indices.to(torch.int32)
    key_ptr = torch.zeros(S + 1, device=col_idx.device, dtype=torch.int32)
    key_ptr[1:] = torch.bincount(col_idx.long(), minlength=S).cumsum(0).to(torch.int32)
    return key_ptr, pair_idx, pair_row


def attn_drop_mask(nnz, seed, p_drop):
    """The keep / scale factors [nnz, 8] the kernels derive from (seed, p_drop) (numpy restatement of the device hash, for tests)."""
    import numpy as np
    if p_drop <= 0:
        return np.ones((nnz, 8), np.float32)
    # exactly as make_drop() (csrc/attention.hip): the kernels receive p_drop as a C float -- thr = (unsigned)((double)(float)p * 2^32),
    # scale = 1.f / (1.f - p) in fp32 (a Python double here would differ by a few counts of the threshold and one ulp of the scale)
    pf = np.float32(p_drop)
    t = float(pf) * 4294967296.0
    thr = (0xffffffff if t >= 4294967295.0 else int(t)) or 1
    with np.errstate(over='ignore'):
        u = (np.arange(nnz * 8, dtype=np.uint64) * np.uint64(0x9E3779B1)).astype(np.uint32) ^ np.uint32(seed & 0xffffffff)
        u ^= u >> np.uint32(16); u = (u.astype(np.uint64) * np.uint64(0x85EBCA6B)).astype(np.uint32)
        u ^= u >> np.uint32(13); u = (u.astype(np.uint64) * np.uint64(0xC2B2AE35)).astype(np.uint32)
        u ^= u >> np.uint32(16)
    return np.where(u >= np.uint32(thr), np.float32(1.0) / (np.float32(1.0) - pf), np.float32(0)).reshape(nnz, 8)


def sparse_xattn_bwd(q, K, V, row_ptr, col_idx, ctx, dctx, R=None, transposed=None, p_drop=0.0, seed=0):
    """Backward of sparse_xattn: returns (dq [R,256] fp32 w.r.t. the pre-scaled q, dK, dV [S,256] fp32); transposed = csr_transpose(...)
    of the same CSR when the caller keeps it."""
    _req(q, torch.float32, 'q'); _req(K, BF16, 'K'); _req(V, BF16, 'V'); _req(ctx, torch.float32, 'ctx'); _req(dctx, torch.float32, 'dctx')
    R = q.shape[0] if R is None else R
    S = K.shape[0]
    nnz = col_idx.numel()
    key_ptr, pair_idx, pair_row = transposed if transposed is not None else csr_transpose(row_ptr, col_idx[:nnz], S)
    dq = torch.empty((R, 256), device=q.device, dtype=torch.float32)
    dK = torch.empty((S, 256), device=q.device, dtype=torch.float32)
    dV = torch.empty((S, 256), device=q.device, dtype=torch.float32)
    pair_ws = torch.empty((max(nnz, 1), 16), device=q.device, dtype=torch.float32)
    # hundreds of keys per query and of queries per key (the self attention over [denoising | matched] queries): the long-row kernels
    long_rows = 1 if (nnz >= 64 * max(R, 1) and nnz >= 64 * max(S, 1)) else 0
    check(_lib.load().mv2d_sparse_xattn_bwd_ex(_p(q), _p(K), _p(V), _p(row_ptr), _p(col_idx), _p(ctx), _p(dctx.contiguous()), _p(key_ptr), _p(pair_idx),
                                               _p(pair_row), _p(pair_ws), _p(dq), _p(dK), _p(dV), R, S, float(p_drop), int(seed) & 0xffffffff, long_rows,
                                               1.0, _stream()), 'mv2d_sparse_xattn_bwd')
    return dq, dK, dV


class SparseCrossAttention(torch.autograd.Function):
    """ctx = softmax over the allowed keys (q . K^T) . V per head; q [R,256] fp32 pre-scaled, K / V [S,256] bf16, CSR (row_ptr, col_idx).
    Differentiable in q, K, V (dK / dV returned in the dtype of their inputs; fp32 K / V are rounded to bf16 for the kernels): the attention core of PETRMultiheadAttention for the
    training path of the head.  transposed = csr_transpose(row_ptr, col_idx, S): the decoder layers share
    one CSR, so the caller builds it once for all of them (otherwise it is built in every backward)."""

    @staticmethod
    def forward(fctx, q, K, V, row_ptr, col_idx, empty_nan=False, transposed=None, p_drop=0.0, seed=0):
        # fp32 K / V are rounded to bf16 here (what the kernels read) and get fp32 gradients back
        # p_drop / seed: attention-probability dropout (training); the backward regenerates the mask from the same pair
        fctx.kv_dtypes = (K.dtype, V.dtype)
        q, K, V = q.contiguous(), K.to(BF16).contiguous(), V.to(BF16).contiguous()
        out = sparse_xattn(q, K, V, row_ptr, col_idx, R=q.shape[0], empty_nan=empty_nan, p_drop=p_drop, seed=seed)
        fctx.save_for_backward(q, K, V, row_ptr, col_idx, out)
        fctx.transposed = transposed
        fctx.drop = (float(p_drop), int(seed))
        return out

    @staticmethod
    def backward(fctx, dout):
        q, K, V, row_ptr, col_idx, out = fctx.saved_tensors
        dq, dK, dV = sparse_xattn_bwd(q, K, V, row_ptr, col_idx, out, dout.float().contiguous(), transposed=fctx.transposed, p_drop=fctx.drop[0],
                                      seed=fctx.drop[1])
        return dq, dK.to(fctx.kv_dtypes[0]), dV.to(fctx.kv_dtypes[1]), None, None, None, None, None, None


def box_params(rois, viewK, viewE, intr, ld_intr, minv, K_roi=None, roi_size=7.0, intr_scale=0.1, min_size=4.0):
    _req(rois, torch.float32, 'rois'); _req(viewK, torch.float64, 'viewK'); _req(viewE, torch.float64, 'viewE')
    check(_lib.load().mv2d_box_params(_p(rois), _p(viewK), _p(viewE), _p(K_roi), _p(intr), ld_intr, _p(minv), rois.shape[0],
                                      roi_size, intr_scale, min_size, _stream()), 'mv2d_box_params')


def refpoint_posemb(center_pred, ld_cp, minv, dim_t, xyz, ref, posemb, R, pc_range_host):
    check(_lib.load().mv2d_refpoint_posemb(_p(center_pred), ld_cp, _p(minv), _p(dim_t), _p(xyz), _p(ref), _p(posemb), R,
                                           pc_range_host.data_ptr(), _stream()), 'mv2d_refpoint_posemb')


def lidar2img_inverse(K_roi, E, out=None):
    _req(K_roi, torch.float64, 'K_roi'); _req(E, torch.float64, 'E')
    R = K_roi.shape[0]
    if out is None:
        out = torch.empty((R, 16), device=K_roi.device, dtype=torch.float32)
    check(_lib.load().mv2d_lidar2img_inverse(_p(K_roi), _p(E), _p(out), R, _stream()), 'mv2d_lidar2img_inverse')
    return out


def posemb3d(ref, dim_t, out=None):
    _req(ref, torch.float32, 'ref')
    R = ref.shape[0]
    if out is None:
        out = torch.empty((R, 384), device=ref.device, dtype=torch.float32)
    check(_lib.load().mv2d_posemb3d(_p(ref), _p(dim_t), _p(out), R, _stream()), 'mv2d_posemb3d')
    return out


def roi_align(map0, rois, H, W, *, map1=None, out0=None, out1=None, out0_f32=None, out1_f32=None, spatial_scale=1.0 / 16,
              sampling_ratio=-1, map1_index=None, out1_is_sum=False, R=None, out0_lo=None, out1_lo=None, out0_lo8=None, out1_lo8=None, lo8_flag=None):
    _req(map0, torch.float32, 'map0'); _req(map1, torch.float32, 'map1'); _req(rois, torch.float32, 'rois')
    _req16(out0, 'out0'); _req16(out1, 'out1'); _req16(out0_lo, 'out0_lo'); _req16(out1_lo, 'out1_lo')
    _req(out0_lo8, torch.uint8, 'out0_lo8'); _req(out1_lo8, torch.uint8, 'out1_lo8'); _req(lo8_flag, torch.int32, 'lo8_flag')
    check(_lib.load().mv2d_roi_align_ex(_p(map0), _p(map1), _p(rois), _p(out0), _p(out1), _p(out0_f32), _p(out1_f32),
                                        rois.shape[0] if R is None else R, H, W, map0.shape[-1], spatial_scale, sampling_ratio,
                                        _p(map1_index), 1 if out1_is_sum else 0, _p(out0_lo), _p(out1_lo), _p(out0_lo8), _p(out1_lo8), _p(lo8_flag), _stream()),
          'mv2d_roi_align')


def box_correlation(rois, view_start, trans, lin, depths, match, V, topk, pad_h, pad_w, max_per_view, sample_size=4,
                    num_depth=8, depth_start=0.5, iou_thr=0.0, ratio=0.0):
    _req(rois, torch.float32, 'rois'); _req(view_start, torch.int32, 'view_start'); _req(trans, torch.float64, 'trans')
    _req(match, torch.int32, 'match')
    check(_lib.load().mv2d_box_correlation(_p(rois), _p(view_start), _p(trans), _p(lin), _p(depths), _p(match), rois.shape[0], V,
                                           sample_size, num_depth, topk, pad_h, pad_w, depth_start, iou_thr, ratio, max_per_view,
                                           _stream()), 'mv2d_box_correlation')


def frame_geometry(rois, viewK, viewE, intr, ld_intr, minv, view_start, trans, lin, depths, match, V, topk, pad_h, pad_w, max_per_view,
                   K_roi=None, roi_size=7.0, intr_scale=0.1, min_size=4.0, sample_size=4, num_depth=8, depth_start=0.5, iou_thr=0.0, ratio=0.0, zero=None):
    """box_params + box_correlation (+ clearing the uint8 buffer `zero`) in one launch."""
    _req(rois, torch.float32, 'rois'); _req(viewK, torch.float64, 'viewK'); _req(viewE, torch.float64, 'viewE'); _req(view_start, torch.int32, 'view_start')
    _req(trans, torch.float64, 'trans'); _req(match, torch.int32, 'match'); _req(zero, torch.uint8, 'zero')
    check(_lib.load().mv2d_frame_geometry(_p(rois), _p(viewK), _p(viewE), _p(K_roi), _p(intr), ld_intr, _p(minv), roi_size, intr_scale, min_size,
                                          _p(view_start), _p(trans), _p(lin), _p(depths), _p(match), rois.shape[0], V, sample_size, num_depth, topk,
                                          pad_h, pad_w, depth_start, iou_thr, ratio, max_per_view, _p(zero), zero.numel() if zero is not None else 0,
                                          _stream()), 'mv2d_frame_geometry')


def csr_workspace_bytes(R, V, h, w):
    return _lib.load().mv2d_csr_workspace_bytes(R, V, h, w)


def mask_compact(rois, match, pad_mask, roi_mask, rect, pos2s, s2pos, S_out, bits_ws, row_count, row_ptr, col_idx, nnz_out,
                 R, V, h, w, topk, stride=16.0, expand_stride=2.0, col_cap=None, n_samples=1):
    """V = views per sample; the maps hold n_samples * V views."""
    col_cap = col_idx.numel() if col_cap is None else col_cap
    check(_lib.load().mv2d_mask_compact(_p(rois), _p(match), _p(pad_mask), _p(roi_mask), _p(rect), _p(pos2s), _p(s2pos), _p(S_out),
                                        _p(bits_ws), _p(row_count), _p(row_ptr), _p(col_idx), _p(nnz_out), col_cap, R, V, h, w, topk,
                                        float(stride), float(expand_stride), n_samples, _stream()), 'mv2d_mask_compact')


def roi_positions(rois, pad_mask, roi_mask, rect, pos2s, s2pos, S_out, R, V, h, w, stride=16.0, expand_stride=1.0):
    check(_lib.load().mv2d_roi_positions(_p(rois), _p(pad_mask), _p(roi_mask), _p(rect), _p(pos2s), _p(s2pos), _p(S_out), R, V, h, w,
                                         float(stride), float(expand_stride), _stream()), 'mv2d_roi_positions')


def roi_positions_csr(rois, pad_mask, roi_mask, rect, pos2s, s2pos, S_out, R, V, h, w, match, row_ptr, col_idx, nnz_out, Vg, topk, stride=16.0,
                      expand_stride=1.0, grp_start=None, order=None, order_flags=None):
    """roi_positions + csr_from_corr (S path) in two launches; order (int32 [R]) + grp_start: also the launch order of the attention blocks."""
    _req(order, torch.int32, 'order'); _req(grp_start, torch.int32, 'grp_start'); _req(order_flags, torch.int32, 'order_flags')
    check(_lib.load().mv2d_roi_positions_csr(_p(rois), _p(pad_mask), _p(roi_mask), _p(rect), _p(pos2s), _p(s2pos), _p(S_out), R, V, h, w, float(stride),
                                             float(expand_stride), _p(match), _p(row_ptr), _p(col_idx), _p(nnz_out), Vg, topk, _p(grp_start),
                                             grp_start.numel() - 1 if grp_start is not None else 0, _p(order), _p(order_flags), _stream()),
          'mv2d_roi_positions_csr')


def csr_from_corr(match, row_ptr, col_idx, nnz_out, R, V, topk):
    check(_lib.load().mv2d_csr_from_corr(_p(match), _p(row_ptr), _p(col_idx), _p(nnz_out), R, V, topk, _stream()),
          'mv2d_csr_from_corr')


def pe_inputs(s2pos, S_dev, S_max, featcl, img2lidar, coords_w, coords_h, coords_d, embeds, dim_t, A_frustum, A_sine, Xf_k16,
              Xf_f32, V, h, w, depth_num, position_range_host, A_frustum_f32=None, A_sine_f32=None):
    """A_frustum [S,3D], A_sine [S,384] (may be None), Xf_k16 [S,256]: key16 rows; *_f32: the same rows unrounded (optional)."""
    _req16(A_frustum, 'A_frustum'); _req16(A_sine, 'A_sine'); _req16(Xf_k16, 'Xf_k16')
    _req(A_frustum_f32, torch.float32, 'A_frustum_f32'); _req(A_sine_f32, torch.float32, 'A_sine_f32')
    check(_lib.load().mv2d_pe_inputs(_p(s2pos), _p(S_dev), S_max, _p(featcl), _p(img2lidar), _p(coords_w), _p(coords_h), _p(coords_d),
                                     _p(embeds), _p(dim_t), _p(A_frustum), _p(A_sine), _p(Xf_k16), _p(Xf_f32), _p(A_frustum_f32),
                                     _p(A_sine_f32), V, h, w, depth_num, position_range_host.data_ptr(), _stream()), 'mv2d_pe_inputs')


def pe_frustum_f32(s2pos, S_dev, S_max, img2lidar, coords_w, coords_h, coords_d, out, V, h, w, depth_num, position_range_host):
    """out [S, 3 D] fp32: the unrounded frustum rows of the PE block at the listed positions (index-exact route; fp64 arithmetic, fast form)."""
    _req(s2pos, torch.int32, 's2pos'); _req(S_dev, torch.int32, 'S_dev'); _req(out, torch.float32, 'out')
    for t, n_ in ((img2lidar, 'img2lidar'), (coords_w, 'coords_w'), (coords_h, 'coords_h'), (coords_d, 'coords_d')):
        _req(t, torch.float64, n_)
    check(_lib.load().mv2d_pe_frustum_f32(_p(s2pos), _p(S_dev), S_max, _p(img2lidar), _p(coords_w), _p(coords_h), _p(coords_d), _p(out), V, h, w,
                                          depth_num, position_range_host.data_ptr(), _stream()), 'mv2d_pe_frustum_f32')
    return out


def decode_topk(cls, reg, R, num_classes, max_num, post_center_range_host, boxes, scores, labels, bbox_index, count, topk_dbg=None,
                grp_start=None, max_grp_rows=0, payload=None):
    """grp_start (int32 [n+1], device) + max_grp_rows: one top-k per sample of a batch, outputs [n][max_num]; payload (optional, fp32
    [n][max_num * 11 + 1]): the wire rows of the all-gather of decoded boxes from the same launch."""
    _req(cls, torch.float32, 'cls'); _req(reg, torch.float32, 'reg'); _req(payload, torch.float32, 'payload')
    n = 0 if grp_start is None else grp_start.numel() - 1
    check(_lib.load().mv2d_decode_topk(_p(cls), _p(reg), R, num_classes, max_num, post_center_range_host.data_ptr(), _p(boxes),
                                       _p(scores), _p(labels), _p(bbox_index), _p(count), _p(topk_dbg), _p(grp_start), n, max_grp_rows,
                                       _p(payload), _stream()), 'mv2d_decode_topk')


def result_pack(boxes, scores, labels, count, score_thr, max_num, out_boxes, out_scores, out_labels, out_count, n_samples=1, in_stride=0):
    """n_samples > 1: inputs [n_samples][in_stride], count [n_samples]; outputs [n_samples][max_num], out_count [n_samples]."""
    check(_lib.load().mv2d_result_pack(_p(boxes), _p(scores), _p(labels), _p(count), float(score_thr), max_num, _p(out_boxes), _p(out_scores),
                                       _p(out_labels), _p(out_count), n_samples, in_stride, _stream()), 'mv2d_result_pack')


def nms_bev(boxes, scores, labels, count, nms_thr, n_samples=1):
    """Rotated BEV NMS per class (nms_thr < 1): returns scores with the suppressed entries at -inf; boxes [n,M,9] or [M,9], count [n] int32."""
    out = torch.empty_like(scores)
    check(_lib.load().mv2d_nms_bev(_p(boxes.contiguous()), _p(scores.contiguous()), _p(labels.contiguous()), _p(count), float(nms_thr), _p(out),
                                   int(n_samples), int(scores.shape[-1]), _stream()), 'mv2d_nms_bev')
    return out


def pack_detections(boxes, scores, labels, count, out, max_num=300):
    """boxes [n,M,9] (or [M,9]), scores, labels int64, count [n] int32 -> out [n, max_num*11 + 1] fp32 (the all-gather payload)."""
    n = count.numel()
    in_stride = scores.shape[-1]
    check(_lib.load().mv2d_pack_detections(_p(boxes), _p(scores), _p(labels), _p(count), _p(out), n, max_num, in_stride, _stream()),
          'mv2d_pack_detections')
    return out


class RoIAlignRows(torch.autograd.Function):
    """RoIAlign (mmcv semantics, 7x7, stride 16) of a position-major fp32 map [rows,256] -> [R,49,256] fp32, differentiable w.r.t. the
    map (``mv2d_roi_align`` / ``mv2d_roi_align_bwd``).  ``index`` (int32 [V*H*W] or None): position -> row of a compacted map (the PE rows
    of the S path); ``full`` is then any full-size map the kernel can read alongside (its output is discarded)."""

    @staticmethod
    def forward(ctx, rows, index, rois, H, W, full=None):
        R = rois.shape[0]
        out = torch.empty((R, 49, 256), device=rows.device, dtype=torch.float32)
        rows_c = rows.contiguous()
        if index is None:
            roi_align(rows_c, rois, H, W, out0_f32=out, R=R)
        else:
            scratch = torch.empty_like(out)
            roi_align(full.contiguous(), rois, H, W, map1=rows_c, out0_f32=scratch, out1_f32=out, map1_index=index, R=R)
        ctx.save_for_backward(rois, index if index is not None else torch.empty(0, dtype=torch.int32, device=rows.device))
        ctx.meta = (H, W, rows.shape[0], index is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        rois, index = ctx.saved_tensors
        H, W, n, indexed = ctx.meta
        gmap = torch.zeros((n, 256), device=gout.device, dtype=torch.float32)
        check(_lib.load().mv2d_roi_align_bwd(_p(gout.contiguous().float()), _p(rois), _p(gmap), _p(index) if indexed else None, rois.shape[0], H, W,
                                             256, 1.0 / 16, -1, _stream()), 'mv2d_roi_align_bwd')
        return gmap, None, None, None, None, None


def match_cost(cls, box, gt, gt_labels, cls_weight=2.0, reg_weight=0.25, alpha=0.25, gamma=2.0):
    """cls [L,R,C] logits, box [L,R,10], gt [G,9] fp32, gt_labels [G] int32 -> cost [L,R,G] fp32 of the Hungarian assignment."""
    L, R, C = cls.shape
    G = gt.shape[0]
    cost = torch.empty(L, R, G, device=cls.device, dtype=torch.float32)
    check(_lib.load().mv2d_match_cost(_p(cls), _p(box), _p(gt), _p(gt_labels), _p(cost), L, R, G, C, cls_weight, reg_weight, alpha, gamma,
                                      _stream()), 'mv2d_match_cost')
    return cost


def set_loss(cls, box, match, gt, gt_labels, code_weights, layer_weights, cls_avg_factor, box_avg_factor, alpha=0.25, gamma=2.0,
             loss_cls_weight=2.0, loss_bbox_weight=0.25, skip_background_boxes=False, need_grad=True):
    """Focal + L1 loss of L layers and its gradient: returns (loss [L,2], dcls [L,R,C] | None, dbox [L,R,10] | None)."""
    L, R, C = cls.shape
    G = gt.shape[0]
    loss = torch.empty(L, 2, device=cls.device, dtype=torch.float32)
    dcls = torch.empty_like(cls) if need_grad else None
    dbox = torch.empty_like(box) if need_grad else None
    check(_lib.load().mv2d_set_loss(_p(cls), _p(box), _p(match), _p(gt), _p(gt_labels), _p(code_weights),
                                    _p(layer_weights) if layer_weights is not None else None, _p(loss),
                                    _p(dcls) if need_grad else None, _p(dbox) if need_grad else None, L, R, G, C, float(cls_avg_factor),
                                    float(box_avg_factor), alpha, gamma, loss_cls_weight, loss_bbox_weight, int(skip_background_boxes),
                                    _stream()), 'mv2d_set_loss')
    return loss, dcls, dbox


SCALE_Q = 1.0 / math.sqrt(32.0)
