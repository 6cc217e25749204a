"""Kernel-level parity: every C-ABI entry point against the oracle / a plain fp32-fp64 torch statement of the
same op, on the GPU (-m gpu).  Integer / boolean results bit-exact; fp32 kernels 1e-5 rel-to-max; bf16 MFMA
kernels compared with an fp64 reference evaluated on the SAME bf16-rounded operands (tolerance 2e-3 from
accumulation order + output rounding)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mv2d_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    from mv2d_amd import _lib
    _lib.load()
    return torch.device('cuda:0')


def relerr(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rnd(shape, seed, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


# ------------------------------------------------------------------------------------------ GEMMs
@pytest.mark.parametrize('M,N,K', [(300, 256, 192), (1000, 1024, 384), (129, 128, 64), (2500, 256, 1024)])
def test_gemm_bf16_plain(dev, M, N, K):
    from mv2d_amd import ops
    A = rnd((M, K), 1).to(dev).to(torch.bfloat16)
    W = rnd((N, K), 2, 0.1).to(dev).to(torch.bfloat16)
    b = rnd((N,), 3).to(dev)
    ref = A.double() @ W.double().T + b.double()
    out = ops.gemm_bf16(A, W, b, out_dtype=torch.float32)
    assert relerr(out, ref) < 1e-4                      # asymmetric operands: a transposed C-write would fail
    out_r = ops.gemm_bf16(A, W, b, act=1)               # relu + bf16 output
    assert relerr(out_r.float(), ref.clamp_min(0)) < 5e-3
    out_s = ops.gemm_bf16(A, W, b, act=2, out_dtype=torch.float32)
    assert relerr(out_s, torch.sigmoid(ref)) < 1e-4


def test_gemm_bf16_epilogues_and_layout(dev):
    from mv2d_amd import ops
    M, K, L = 700, 256, 3
    Xk = rnd((M, K), 4).to(dev).to(torch.bfloat16)
    Xv = rnd((M, K), 5).to(dev).to(torch.bfloat16)
    W = rnd((2 * L * 256, K), 6, 0.1).to(dev).to(torch.bfloat16)
    b = rnd((2 * L * 256,), 7).to(dev)
    m_dev = torch.tensor([650], dtype=torch.int32, device=dev)
    out = torch.zeros((2 * L, 800, 256), device=dev, dtype=torch.bfloat16)          # [K layers | V layers][S_max][256]
    ops.gemm_bf16(Xk, W, b, A2=Xv, n_split=L * 256, m_dev=m_dev, out=out, ldc=256, c_blk_stride=800 * 256, c_blk_cols=256)
    refk = (Xk.double() @ W[:L * 256].double().T + b[:L * 256].double())[:650]
    refv = (Xv.double() @ W[L * 256:].double().T + b[L * 256:].double())[:650]
    for l in range(L):
        assert relerr(out[l, :650].float(), refk[:, l * 256:(l + 1) * 256]) < 5e-3
        assert relerr(out[L + l, :650].float(), refv[:, l * 256:(l + 1) * 256]) < 5e-3
    assert float(out[:, 650:].float().abs().max()) == 0.0                           # rows >= *m_dev untouched
    # mul / add / second output
    mul = rnd((M, 256), 8).to(dev)
    add = rnd((M, 256), 9).to(dev)
    add2 = rnd((M, 256), 10).to(dev)
    W1 = W[:256].contiguous()
    out1 = torch.empty((M, 256), device=dev, dtype=torch.float32)
    out2 = torch.empty((M, 256), device=dev, dtype=torch.bfloat16)
    ops.gemm_bf16(Xk, W1, b[:256].contiguous(), mul=mul, add=add, out=out1, out2=out2, add2=add2)
    ref = (Xk.double() @ W1.double().T + b[:256].double()) * mul.double() + add.double()
    assert relerr(out1, ref) < 1e-4
    assert relerr(out2.float(), ref + add2.double()) < 5e-3


def test_gemm_bf16_conv3x3(dev):
    from mv2d_amd import ops
    R = 37
    x = rnd((R, 256, 7, 7), 11).to(torch.bfloat16)
    w = rnd((256, 256, 3, 3), 12, 0.05).to(torch.bfloat16)
    b = rnd((256,), 13)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))            # [R,256,7,7]
    x_cl = x.flatten(2).transpose(1, 2).contiguous().to(dev)                         # [R,49,256]
    w_r = w.permute(0, 2, 3, 1).reshape(256, 9 * 256).contiguous().to(dev)           # [out][tap][cin]
    out = ops.gemm_bf16(x_cl, w_r, b.to(dev), conv3x3=True, act=1, out_dtype=torch.float32)
    assert relerr(out.view(R, 49, 256).transpose(1, 2), ref.flatten(2)) < 1e-4


@pytest.mark.parametrize('M,N,K,split', [(300, 256, 256, 1), (300, 768, 256, 1), (77, 10, 256, 1), (300, 256, 2048, 8), (900, 2048, 256, 1), (50, 512, 1056, 1)])
def test_gemm_f32_exact(dev, M, N, K, split):
    from mv2d_amd import ops
    A = rnd((M, K), 20).to(dev)
    W = rnd((N, K), 21, 0.1).to(dev)
    b = rnd((N,), 22).to(dev)
    ref = A.double() @ W.double().T + b.double()
    out = ops.gemm_f32(A, W, b, split_k=split)
    if split > 1:
        out = out.sum(0)
    assert relerr(out, ref) < 2e-6
    out2 = ops.gemm_f32(A, W, b, act=1, scale=0.25, out_dtype=torch.bfloat16) if split == 1 else None
    if out2 is not None:
        assert relerr(out2.float(), (ref * 0.25).clamp_min(0)) < 5e-3


def test_gemm_f32_a_select(dev):
    from mv2d_amd import ops
    A = rnd((300, 256), 23).to(dev)
    A2 = rnd((300, 256), 24).to(dev)
    W = rnd((768, 256), 25, 0.1).to(dev)
    b = rnd((768,), 26).to(dev)
    out = ops.gemm_f32(A, W, b, A2=A2, n_split=512)
    ref = torch.cat([A.double() @ W[:512].double().T, A2.double() @ W[512:].double().T], 1) + b.double()
    assert relerr(out, ref) < 2e-6


def test_attn_out_fused(dev):
    from mv2d_amd import ops
    for M in (300, 21):
        ctx = rnd((M, 256), 41).to(dev); res = rnd((M, 256), 42).to(dev); qpos = rnd((M, 256), 43).to(dev)
        Wo = rnd((256, 256), 44, 0.1).to(dev); bo = rnd((256,), 45).to(dev); Wq = rnd((256, 256), 46, 0.1).to(dev); bq = rnd((256,), 47).to(dev)
        lw = rnd((256,), 48).to(dev); lb = rnd((256,), 49).to(dev)
        x1 = torch.empty((M, 256), device=dev); q = torch.empty((M, 256), device=dev)
        ops.attn_out_fused(ctx, res, Wo, bo, (lw, lb), x1, qpos=qpos, Wq=Wq, bq=bq, qscale=0.25, q_out=q)
        r1 = F.layer_norm(ctx.double() @ Wo.double().T + bo.double() + res.double(), (256,), lw.double(), lb.double())
        assert relerr(x1, r1) < 1e-5
        assert relerr(q, ((r1 + qpos.double()) @ Wq.double().T + bq.double()) * 0.25) < 1e-5
        x2 = torch.empty((M, 256), device=dev)
        ops.attn_out_fused(ctx, res, Wo, bo, (lw, lb), x2)
        assert torch.equal(x1, x2)


def test_heads_fused(dev):
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    sd = synthetic.make_head_state(seed=0)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    L, M = 6, 77
    outs = rnd((L, M, 256), 51)
    ref = torch.from_numpy(np.random.Generator(np.random.PCG64(52)).random((M, 3)).astype(np.float32)) * 1.4 - 0.2
    cls_ref, reg_ref = O.pred_heads(sdt, outs, ref)
    reg_ref = torch.cat([reg_ref[..., :8], reg_ref[..., 8:] / 0.5], -1)
    st = lambda fmt: torch.stack([sdt[fmt.format(l)] for l in range(L)]).contiguous().to(dev)
    cw = [st('bbox_head.cls_branches.{}.' + n) for n in ('0.weight', '0.bias', '1.weight', '1.bias', '3.weight', '3.bias', '4.weight', '4.bias', '6.weight', '6.bias')]
    rw = [st('bbox_head.reg_branches.{}.' + n) for n in ('0.weight', '0.bias', '2.weight', '2.bias', '4.weight', '4.bias')]
    for i in (0, 4):
        cw[i] = ops.pack_wfrag_f32(cw[i])            # the 256x256 matrices go in fragment-major
    for i in (0, 2):
        rw[i] = ops.pack_wfrag_f32(rw[i])
    cls = torch.empty((L, M, 10), device=dev); reg = torch.empty((L, M, 10), device=dev)
    ops.heads_fused(outs.to(dev), ops.make_ptr_array(cw), ops.make_ptr_array(rw), ref.to(dev), cls, reg, M, L,
                    torch.tensor(O.PC_RANGE, dtype=torch.float32), dt=0.5)
    assert relerr(cls, cls_ref) < 1e-5
    assert relerr(reg, reg_ref) < 1e-5


@pytest.mark.parametrize('M', [77, 531])            # one / two row tiles per block
def test_heads_fused_x3(dev, M):
    """prediction branches with the 256x256 linears in bf16x3: fp32-class agreement with the oracle, per-row dt of a batch"""
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    sd = synthetic.make_head_state(seed=0)
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    L = 6
    outs = rnd((L, M, 256), 51)
    ref = torch.from_numpy(np.random.Generator(np.random.PCG64(52)).random((M, 3)).astype(np.float32)) * 1.4 - 0.2
    cls_ref, reg_ref = O.pred_heads(sdt, outs, ref)
    dt_rows = torch.where(torch.arange(M) < 40, 0.5, 0.25).float()
    reg_ref = torch.cat([reg_ref[..., :8], reg_ref[..., 8:] / dt_rows[None, :, None]], -1)
    st = lambda fmt: torch.stack([sdt[fmt.format(l)] for l in range(L)]).contiguous().to(dev)
    c = {n: st('bbox_head.cls_branches.{}.' + n) for n in ('0.weight', '0.bias', '1.weight', '1.bias', '3.weight', '3.bias', '4.weight', '4.bias', '6.weight', '6.bias')}
    r = {n: st('bbox_head.reg_branches.{}.' + n) for n in ('0.weight', '0.bias', '2.weight', '2.bias', '4.weight', '4.bias')}
    cw = [*ops.pack_x3_stack(c['0.weight']), c['0.bias'], c['1.weight'], c['1.bias'], *ops.pack_x3_stack(c['3.weight']), c['3.bias'], c['4.weight'],
          c['4.bias'], c['6.weight'], c['6.bias']]
    rw = [*ops.pack_x3_stack(r['0.weight']), r['0.bias'], *ops.pack_x3_stack(r['2.weight']), r['2.bias'], r['4.weight'], r['4.bias']]
    cls = torch.empty((L, M, 10), device=dev); reg = torch.empty((L, M, 10), device=dev)
    ops.heads_fused_x3(outs.to(dev), ops.make_ptr_array(cw), ops.make_ptr_array(rw), ref.to(dev), cls, reg, M, L,
                       torch.tensor(O.PC_RANGE, dtype=torch.float32), dt=123.0, dt_rows=dt_rows.to(dev))
    assert relerr(cls, cls_ref) < 5e-5
    assert relerr(reg, reg_ref) < 5e-5


@pytest.mark.parametrize('M,N,K', [(300, 1024, 256), (1200, 512, 1056), (77, 256, 512), (531, 768, 256), (1, 16, 32)])
def test_linear_x3(dev, M, N, K):
    """LDS-tiled bf16x3 linear: fp32-class agreement with fp64, ReLU / clamp epilogue, leading dimensions, ragged edges, two inputs"""
    from mv2d_amd import ops
    A = rnd((M, K), 61).to(dev); A2 = rnd((M, K), 62).to(dev)
    W = rnd((N, K), 63, 0.1).to(dev); b = rnd((N,), 64).to(dev)
    Wx = ops.pack_x3(W)
    ref = A.double() @ W.double().T + b.double()
    out = ops.linear_x3(A, Wx, b, N=N, K=K)
    assert relerr(out, ref) < 3e-5
    big = torch.full((M, N + 40), 7.0, device=dev)
    ops.linear_x3(A, Wx, b, N=N, K=K, act=1, clamp=1.5, out=big, ldc=N + 40)
    err = float((big[:, :N].double() - ref.relu().clamp(max=1.5)).abs().max() / ref.abs().max())       # error relative to the pre-clamp scale
    assert err < 3e-5 and bool((big[:, N:] == 7.0).all())
    if N % 256 == 0:
        out2 = ops.linear_x3(A, Wx, b, N=N, K=K, A2=A2, n_split=N // 2)
        ref2 = torch.cat([ref[:, :N // 2], (A2.double() @ W.double().T + b.double())[:, N // 2:]], 1)
        assert relerr(out2, ref2) < 3e-5


def test_ffn_fused_exact(dev):
    from mv2d_amd import ops
    for M in (300, 33, 900):
        x = rnd((M, 256), 27).to(dev)
        W1 = rnd((2048, 256), 28, 0.1).to(dev); b1 = rnd((2048,), 29).to(dev)
        W2 = rnd((256, 2048), 30, 0.05).to(dev)
        slabs = ops.ffn_fused(x, *ops.ffn_pack_weights(W1, W2)[:1], b1, ops.ffn_pack_weights(W1, W2)[1])
        assert slabs.shape == (32, M, 256)
        ref = F.relu(x.double() @ W1.double().T + b1.double()) @ W2.double().T
        assert relerr(slabs.sum(0), ref) < 2e-6


def test_ffn_fused_x3_split_precision(dev):
    """bf16x3 FFN: fp32-class accuracy (<= 3e-5 of the output maximum) against fp64, same slab contract as the exact kernel."""
    from mv2d_amd import ops
    for M in (300, 33, 900):
        x = rnd((M, 256), 27).to(dev)
        W1 = rnd((2048, 256), 28, 0.1).to(dev); b1 = rnd((2048,), 29).to(dev)
        W2 = rnd((256, 2048), 30, 0.05).to(dev)
        slabs = ops.ffn_fused_x3(x, ops.pack_x3(W1), b1, ops.pack_x3(W2))
        assert slabs.shape == (32, M, 256)
        ref = F.relu(x.double() @ W1.double().T + b1.double()) @ W2.double().T
        assert relerr(slabs.sum(0), ref) < 3e-5
        W1p, W2p = ops.ffn_pack_weights(W1, W2)
        exact = ops.ffn_fused(x, W1p, b1, W2p)
        assert relerr(slabs.sum(0), exact.sum(0)) < 3e-5
        for G in (2, 4, 8):                                   # G hidden slices accumulated per block: 32 / G slabs
            sg = ops.ffn_fused_x3(x, ops.pack_x3(W1), b1, ops.pack_x3(W2), groups=G)
            assert sg.shape == (32 // G, M, 256) and relerr(sg.sum(0), ref) < 3e-5


# ------------------------------------------------------------------------------------------ rows
def test_row_ln(dev):
    from mv2d_amd import ops
    M = 301
    parts = rnd((8, M, 256), 30).to(dev)
    bias = rnd((256,), 31).to(dev)
    res = rnd((M, 256), 32).to(dev)
    w1, b1, w2, b2 = [rnd((256,), 33 + i).to(dev) for i in range(4)]
    qpos = rnd((M, 256), 37).to(dev)
    out = torch.empty((M, 256), device=dev)
    outp = torch.empty((M, 256), device=dev)
    out2 = torch.empty((M, 256), device=dev)
    ops.row_ln(parts, bias=bias, residual=res, ln=(w1, b1), out=out, addvec=qpos, out_plus=outp, ln2=(w2, b2), out2=out2)
    v = parts.double().sum(0) + bias.double() + res.double()
    y = F.layer_norm(v, (256,), w1.double(), b1.double())
    assert relerr(out, y) < 1e-5
    assert relerr(outp, y + qpos.double()) < 1e-5
    assert relerr(out2, F.layer_norm(y, (256,), w2.double(), b2.double())) < 1e-5
    o3 = ops.row_ln(parts[0].contiguous(), ln=(w1, b1), relu=True)
    assert relerr(o3, F.relu(F.layer_norm(parts[0].double(), (256,), w1.double(), b1.double()))) < 1e-5


def test_transpose_and_cast(dev):
    from mv2d_amd import ops
    x = rnd((3, 256, 5, 7), 40).to(dev)
    y = ops.nchw_to_nhwc(x)
    assert torch.equal(y.view(3, 5, 7, 256), x.permute(0, 2, 3, 1))
    for shape in ((2, 256, 32, 88), (2, 256, 14, 26), (1, 256, 9, 12)):          # 16-byte path: full / ragged tiles
        x2 = rnd(shape, 41).to(dev)
        assert torch.equal(ops.nchw_to_nhwc(x2).view(shape[0], shape[2], shape[3], 256), x2.permute(0, 2, 3, 1))
    z = ops.f32_to_bf16(x)
    assert torch.equal(z, x.to(torch.bfloat16))


def test_key16_conversion_rounds_to_nearest_and_saturates(dev):
    """The key-side 16-bit format (csrc/common.h key16 = fp16 since round 4): round-to-nearest-even, SATURATING at +-65504 (an fp16 inf in a
    key row would poison a softmax row; a saturated element is finite), NaN kept; hi + lo pairs carry ~22 bits down to an absolute 2^-24."""
    from mv2d_amd import ops
    k16 = ops.key16_dtype()
    x = torch.cat([rnd((4099,), 43) * 3.0, torch.tensor([0., 1., -1., 1e-3, 3e-8, 6e-5, 2049.0, 2051.0, 65504., 65519.9, 65520., 7e4, -1e9, float('inf'), float('-inf')])]).to(dev)
    hi = ops.f32_to_key16(x)
    hi2, lo = ops.f32_to_key16(x, with_lo=True)
    assert torch.equal(hi, hi2)
    if k16 == torch.float16:
        assert torch.equal(hi, x.clamp(-65504.0, 65504.0).to(k16))
        assert bool(torch.isfinite(hi.float()).all())
        fin = x.abs() < 65504.0
        err = (hi.double() + lo.double() - x.double()).abs()[fin]
        assert float((err - (x.double().abs()[fin] * 2.0 ** -21)).clamp_min(0).max()) <= 2.0 ** -24      # 2^-22-class relative, 2^-25 absolute floor
        assert float(lo[~fin].float().abs().max()) == 0.0                                                    # a saturated value has no remainder
    else:
        assert torch.equal(hi, x.to(k16))
    n = ops.f32_to_key16(torch.tensor([float('nan'), 1.0], device=dev))
    assert bool(torch.isnan(n[0])) and float(n[1]) == 1.0


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize('R', [12, 300, 333])
def test_self_attn(dev, R):
    from mv2d_amd import ops
    qkv = rnd((R, 768), 50).to(dev)
    qkv[:, :256] *= 3.0                                   # sharper logits: the running maximum moves between the 32-key steps
    q, k, v = [t.double().view(R, 8, 32).transpose(0, 1) for t in qkv.split(256, 1)]
    att = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(32), -1)
    ref = (att @ v).transpose(0, 1).reshape(R, 256)
    for impl in ('x3', 'f32'):                            # default: bf16 split precision through LDS; f32: the round-1 exact-fp32 kernel
        out = ops.self_attn(qkv, impl=impl)
        # split precision drops the lo x lo terms (2^-18 of |q||k|): with logits of magnitude ~15 that is 5e-5 after the exponential
        assert relerr(out, ref) < (6e-5 if impl == 'x3' else 1e-5), (impl, relerr(out, ref))
    qkv[:, :256] /= 3.0                                   # logits of the size the decoder produces: fp32-class either way
    q = qkv[:, :256].double().view(R, 8, 32).transpose(0, 1)
    ref = (torch.softmax(q @ k.transpose(1, 2) / math.sqrt(32), -1) @ v).transpose(0, 1).reshape(R, 256)
    assert relerr(ops.self_attn(qkv, impl='x3'), ref) < 1e-5


@pytest.mark.parametrize('sizes', [(300, 300, 300), (37, 1, 290, 64, 129), (900, 450)])
def test_self_attn_samples_of_a_batch(dev, sizes):
    """Several samples in one launch (grp_start): attention stays inside a sample and a sample's rows are BITWISE what a single-sample
    launch gives, also with padding rows behind the last sample (the engine's RoI buckets) and a loose per-sample row bound."""
    from mv2d_amd import ops
    R = sum(sizes)
    pad = 40
    qkv = rnd((R + pad, 768), 51).to(dev)
    grp = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    for impl in ('x3', 'f32'):
        out = ops.self_attn(qkv, grp_start=grp, max_grp_rows=max(sizes) + 23 if impl == 'x3' else 0, impl=impl)
        o = 0
        for n in sizes:
            single = ops.self_attn(qkv[o:o + n].contiguous(), impl=impl)
            assert torch.equal(out[o:o + n], single), (impl, n)
            o += n


def test_sparse_xattn(dev):
    from mv2d_amd import ops
    R, S = 301, 5000
    g = np.random.Generator(np.random.PCG64(60))
    allowed = torch.from_numpy(g.random((R, S)) < 0.02)
    allowed[5] = False                                   # a query with no key -> ctx = 0
    allowed[7, :] = False
    allowed[7, 123] = True                               # a single key
    q = rnd((R, 256), 61).to(dev)
    K = rnd((S, 256), 62).to(dev).to(torch.bfloat16)
    V = rnd((S, 256), 63).to(dev).to(torch.bfloat16)
    counts = allowed.sum(1)
    row_ptr = torch.zeros(R + 1, dtype=torch.int32)
    row_ptr[1:] = counts.cumsum(0)
    col = allowed.nonzero()[:, 1].to(torch.int32)
    nnz = int(row_ptr[-1])
    dbg = torch.zeros((8, nnz), device=dev)
    out = ops.sparse_xattn(q, K, V, row_ptr.to(dev), col.to(dev), dbg_logits=dbg, empty_nan=False)
    out_nan = ops.sparse_xattn(q, K, V, row_ptr.to(dev), col.to(dev))      # default: NaN like nn.MultiheadAttention
    assert bool(torch.isnan(out_nan[5]).all()) and torch.equal(out_nan[6:], out[6:]) and torch.equal(out_nan[:5], out[:5])
    qh = q.double().view(R, 8, 32).transpose(0, 1)
    kh = K.double().view(S, 8, 32).transpose(0, 1)
    vh = V.double().view(S, 8, 32).transpose(0, 1)
    logits = qh @ kh.transpose(1, 2)
    lm = logits.masked_fill(~allowed.to(dev)[None], float('-inf'))
    att = torch.softmax(lm, -1).nan_to_num(0.0)
    ref = (att @ vh).transpose(0, 1).reshape(R, 256)
    assert relerr(out, ref) < 1e-5
    assert float(out[5].abs().max()) == 0.0
    lg_ref = logits[:, allowed.to(dev)]                  # [8, nnz] in CSR (row-major) order
    assert relerr(dbg, lg_ref) < 1e-5


@pytest.mark.parametrize('R,S,dens', [(37, 500, 0.05), (301, 5000, 0.02), (64, 49 * 64, -1.0)])
def test_sparse_xattn_backward(dev, R, S, dens):
    """"next" row f3: gradients of the sparse cross attention (dq, dK, dV) == torch autograd of the oracle's dense masked attention in
    fp64 on the same bf16-rounded keys / values; keys shared by several queries, a row without keys, a single-key row; dens < 0 = the
    S-path pattern (every query reads the 49 cells of its own RoI)."""
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    g = np.random.Generator(np.random.PCG64(160 + R))
    if dens > 0:
        allowed = torch.from_numpy(g.random((R, S)) < dens)
        allowed[5] = False                                   # no key at all
        allowed[7, :] = False
        allowed[7, 123] = True                               # a single key
    else:
        allowed = torch.zeros((R, S), dtype=torch.bool)
        for r in range(R):
            allowed[r, r * 49:(r + 1) * 49] = True
            allowed[r, ((r + 3) % R) * 49:((r + 3) % R) * 49 + 49] = True    # one correlated RoI
    q = rnd((R, 256), 161).to(dev)
    K = rnd((S, 256), 162).to(dev).to(torch.bfloat16)
    V = rnd((S, 256), 163).to(dev).to(torch.bfloat16)
    dout = rnd((R, 256), 164).to(dev)
    row_ptr, col = O.csr_from_allowed(allowed)
    qd = q.double().requires_grad_(True); Kd = K.double().requires_grad_(True); Vd = V.double().requires_grad_(True)
    ref = O.masked_cross_attention(qd, Kd, Vd, allowed.to(dev))
    ref.backward(dout.double())
    # through the autograd wrapper of the product path
    q2 = q.clone().requires_grad_(True); K2 = K.clone().requires_grad_(True); V2 = V.clone().requires_grad_(True)
    tr = ops.csr_transpose(row_ptr.to(dev), col.to(dev), S)
    out = ops.SparseCrossAttention.apply(q2, K2, V2, row_ptr.to(dev), col.to(dev), False, tr)
    assert relerr(out, ref) < 1e-5
    out.backward(dout)
    assert relerr(q2.grad, qd.grad) < 1e-5
    dq, dK, dV = ops.sparse_xattn_bwd(q, K, V, row_ptr.to(dev), col.to(dev), out.detach(), dout)       # fp32 gradients
    assert relerr(dq, qd.grad) < 1e-5 and relerr(dK, Kd.grad) < 2e-5 and relerr(dV, Vd.grad) < 2e-5
    assert relerr(K2.grad.float(), Kd.grad) < 8e-3 and relerr(V2.grad.float(), Vd.grad) < 8e-3           # returned in bf16
    d2 = ops.sparse_xattn_bwd(q, K, V, row_ptr.to(dev), col.to(dev), out.detach(), dout)
    assert all(torch.equal(a, b) for a, b in zip((dq, dK, dV), d2))                                    # deterministic (no atomics)
    if dens > 0:
        assert float(dq[5].abs().max()) == 0.0
        unused = ~allowed.any(0)
        assert float(dK[unused.to(dev)].abs().max()) == 0.0 and float(dV[unused.to(dev)].abs().max()) == 0.0


def _unpack_qt(Qt, R):
    """Qt [R,4096] key16 -> (hi [R,8,256], lo [R,8,256]) fp64: Qt[r][h][s][g][part][e] = part (hi | lo) of head h, channel 32 s + 8 g + e."""
    t = Qt.view(R, 8, 8, 4, 2, 8).double().cpu()                    # [r][h][s][g][part][e]
    return t[:, :, :, :, 0].reshape(R, 8, 256), t[:, :, :, :, 1].reshape(R, 8, 256)


@pytest.mark.parametrize('R,S,dens,lo', [(300, 700, 0.2, True), (37, 300, 0.5, True), (64, 64 * 49, 0.0, True), (300, 700, 0.2, False)])
def test_xattn_fused_equals_the_three_kernels(dev, R, S, dens, lo):
    """mv2d_xattn_fused_fwd (csrc/xattn_fused.hip, round 5: query map -> tile attention -> context map per block of 8 queries, Qt / z on chip)
    == the three launches of csrc/xattn_tile.hip with one wave per query, BIT FOR BIT: ragged R (not a multiple of 8), rows of 0 .. 300 keys,
    an empty row (NaN and zero policy), hi-only and hi + lo rows, any launch order."""
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    g = np.random.Generator(np.random.PCG64(460 + R))
    if dens > 0:
        allowed = torch.from_numpy(g.random((R, S)) < dens)
        allowed[5] = False
        allowed[7, :] = False
        allowed[7, 123] = True
    else:
        allowed = torch.zeros((R, S), dtype=torch.bool)
        for r in range(R):
            allowed[r, r * 49:(r + 1) * 49] = True
            if r % 3 == 0:
                allowed[r, ((r + 3) % R) * 49:((r + 3) % R) * 49 + 49] = True
    k16 = ops.key16_dtype()
    q = (rnd((R, 256), 461) * 0.3).to(dev)
    q[3] *= 8.0
    xk32, xv32 = rnd((S, 256), 462).to(dev), rnd((S, 256), 463).to(dev)
    Xk, Xk_lo = ops.f32_to_key16(xk32, with_lo=True)
    Xv, Xv_lo = ops.f32_to_key16(xv32, with_lo=True)
    if not lo:
        Xk_lo = Xv_lo = None
    Wk, Wv = rnd((256, 256), 464, 0.06).to(dev), rnd((256, 256), 466, 0.06).to(dev)
    bv = rnd((256,), 467).to(dev)
    row_ptr, col = O.csr_from_allowed(allowed)
    row_ptr, col = row_ptr.to(dev), col.to(dev)
    WA, WB = ops.pack_xattn_maps(Wk, Wv)
    for empty_nan in (False, True):
        Qt = ops.xattn_qmap(q, WA)
        z = ops.xattn_tile(Qt, Xk, Xv, row_ptr, col, empty_nan=empty_nan, waves=1, Xk_lo=Xk_lo, Xv_lo=Xv_lo)
        ref = ops.xattn_ctxmap(z, WB, bv, row_ptr, empty_nan=empty_nan)
        out = ops.xattn_fused(q, WA, WB, bv, Xk, Xv, row_ptr, col, empty_nan=empty_nan, Xk_lo=Xk_lo, Xv_lo=Xv_lo)
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), (empty_nan, float((out - ref).abs().nan_to_num(0).max()))
        perm = torch.randperm(R, generator=torch.Generator().manual_seed(R)).to(torch.int32).to(dev)
        out2 = ops.xattn_fused(q, WA, WB, bv, Xk, Xv, row_ptr, col, empty_nan=empty_nan, Xk_lo=Xk_lo, Xv_lo=Xv_lo, order=perm)
        assert torch.equal(out2.view(torch.int32), ref.view(torch.int32))
    if dens > 0:
        assert bool(torch.isnan(out[5]).all()) and bool(torch.isfinite(out[6:]).all())
    # a mapped query beyond the fp16 range (saturating splits): finite results; a NaN in a key row, a value row or the query: every row that lists it is NaN
    big = ops.xattn_fused(q * 3e4, WA, WB, bv, Xk, Xv, row_ptr, col, empty_nan=False, Xk_lo=Xk_lo, Xv_lo=Xv_lo)
    assert bool(torch.isfinite(big).all())
    if dens > 0:
        hit = allowed[:, 123].to(dev)
        for which in ('k', 'v', 'q'):
            k2, v2, q2 = Xk.clone(), Xv.clone(), q.clone()
            if which == 'k':
                k2[123] = float('nan')
            elif which == 'v':
                v2[123] = float('nan')
            else:
                q2[9] = float('nan')
            o_ = ops.xattn_fused(q2, WA, WB, bv, k2, v2, row_ptr, col, empty_nan=False, Xk_lo=Xk_lo, Xv_lo=Xv_lo)
            bad = torch.isnan(o_).all(1)
            want = hit if which != 'q' else (torch.arange(R, device=dev) == 9) & (row_ptr[1:] > row_ptr[:-1])
            assert torch.equal(bad, want), which
    if lo:
        # round 6, e4m3 "lo8" rows (256-byte lo rows, csrc/common.h): the kernels decode the bytes to key16 in registers -- BITWISE the results of key16 lo
        # rows that hold the decoded values, for the tile kernel (1, 2, 4 waves per query) and the fused one; and close to the unquantised rows
        k8, v8 = ops.lo8_encode(Xk_lo), ops.lo8_encode(Xv_lo)
        kd, vd = ops.lo8_decode(k8), ops.lo8_decode(v8)
        assert float((kd.float() - Xk_lo.float()).abs().max()) <= float(Xk_lo.float().abs().max()) * 2.0 ** -4
        Qt = ops.xattn_qmap(q, WA)
        for waves in (1, 2, 4):
            z8 = ops.xattn_tile(Qt, Xk, Xv, row_ptr, col, waves=waves, Xk_lo=k8, Xv_lo=v8)
            zd = ops.xattn_tile(Qt, Xk, Xv, row_ptr, col, waves=waves, Xk_lo=kd, Xv_lo=vd)
            assert torch.equal(z8.view(torch.int32), zd.view(torch.int32)), waves
        o8 = ops.xattn_fused(q, WA, WB, bv, Xk, Xv, row_ptr, col, Xk_lo=k8, Xv_lo=v8)
        od = ops.xattn_fused(q, WA, WB, bv, Xk, Xv, row_ptr, col, Xk_lo=kd, Xv_lo=vd)
        assert torch.equal(o8.view(torch.int32), od.view(torch.int32))
        assert torch.equal(o8.view(torch.int32), ops.xattn_ctxmap(ops.xattn_tile(Qt, Xk, Xv, row_ptr, col, waves=1, Xk_lo=k8, Xv_lo=v8), WB, bv, row_ptr).view(torch.int32))
        fin = torch.isfinite(out)
        # (against the unquantised lo rows: the logits of this test reach +-100 -- q[3] is scaled by 8 -- so their 2^-16 relative change is visible in the softmax;
        #  the head's own logits are what tests/test_gpu_golden.py measures: class logits within 1.1e-6 of the reference's)
        assert float((o8 - out)[fin].abs().max()) < 1e-3 * float(out[fin].abs().max())
        with pytest.raises(Exception):
            ops.xattn_fused(q, WA, WB, bv, Xk, Xv, row_ptr, col, Xk_lo=k8, Xv_lo=Xv_lo)          # mixed lo formats


def test_lo8_row_format(dev):
    """The e4m3 "lo8" conversions of csrc/common.h against torch's OCP e4m3: every byte decodes to fp16(e4m3 / 2^12) (subnormal results kept, the
    two NaN bytes stay NaN), and the row producers (RoIAlign outputs; the PE kernels: test_pe_fused_x3_kernel) write round-to-nearest-even bytes of
    their key16 lo halves, saturating at +-448."""
    from mv2d_amd import ops
    k16 = ops.key16_dtype()
    if k16 != torch.float16:
        pytest.skip('lo8 rows need the fp16 key format')
    # decode through the attention kernel: one query, one head-map = identity is more plumbing than it is worth; the producers + ops.lo8_decode pin the
    # format, tools/probes/fp8_probe.hip checks the two instructions exhaustively.  Here: RoIAlign's lo8 outputs == lo8_encode of its key16 lo outputs.
    H, W, R = 24, 40, 37
    g = torch.Generator().manual_seed(77)
    m0 = (torch.randn((2 * H * W, 256), generator=g) * torch.logspace(-3, 2.6, 256)).to(dev)       # magnitudes 1e-3 .. 400: subnormal and saturated lo bytes
    m1 = torch.randn((2 * H * W, 256), generator=g).to(dev)
    x1 = torch.rand(R, generator=g) * (W * 16 - 64); y1 = torch.rand(R, generator=g) * (H * 16 - 64)
    rois = torch.stack([torch.randint(0, 2, (R,), generator=g).float(), x1, y1, x1 + 8 + torch.rand(R, generator=g) * 120, y1 + 8 + torch.rand(R, generator=g) * 120], 1).to(dev)
    o = {n_: torch.zeros((R, 49, 256), device=dev, dtype=k16) for n_ in ('a', 'a_lo', 'b', 'b_lo', 'a2', 'b2', 'a2_lo')}
    o8 = {n_: torch.zeros((R, 49, 256), device=dev, dtype=torch.uint8) for n_ in ('a', 'b')}
    ops.roi_align(m0, rois, H, W, map1=m1, out0=o['a'], out1=o['b'], out1_is_sum=True, out0_lo=o['a_lo'], out1_lo=o['b_lo'])
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.roi_align(m0, rois, H, W, map1=m1, out0=o['a2'], out1=o['b2'], out1_is_sum=True, out0_lo=o['a2_lo'], out0_lo8=o8['a'], out1_lo8=o8['b'], lo8_flag=flag)
    torch.cuda.synchronize()
    assert torch.equal(o['a'], o['a2']) and torch.equal(o['b'], o['b2']) and torch.equal(o['a_lo'], o['a2_lo'])
    for n_ in ('a', 'b'):
        want = ops.lo8_encode(o[n_ + '_lo'])
        assert torch.equal(o8[n_], want), int((o8[n_] != want).sum())
        dec = ops.lo8_decode(o8[n_]).float()
        lo = o[n_ + '_lo'].float()
        # relative 2^-4 in the normal range of the format, absolute 2^-22 below it; the largest magnitudes saturate (|lo| * 2^12 > 448)
        sat = lo.abs() * 4096.0 > 448.0
        assert bool((((dec - lo).abs() <= lo.abs() * 2.0 ** -4 + 2.0 ** -22) | sat).all())
        assert int(((o8[n_] & 0x7f) < 8).sum()) > 0                                    # subnormal bytes occur
    assert bool((ops.lo8_decode(o8['a'])[(o['a_lo'].float().abs() * 4096.0 > 460.0)].float().abs() == 448.0 / 4096.0).all())
    # the saturation report: set by this map (values up to 400), not by one inside the format's range
    assert int(flag.item()) == 1
    flag.zero_()
    ops.roi_align(m1, rois, H, W, out0=o['a2'], out0_lo8=o8['a'], lo8_flag=flag)
    assert int(flag.item()) == 0


def _rect_pattern(R, S, seed, n_samples):
    """A T-path-like mask: every query lists 1-3 runs of keys inside its sample's key range, runs shared between neighbouring queries
    (the structure of RH/mv2d_t_head.py:84-88: rectangles of correlated RoIs), one empty row, one single-key row."""
    g = np.random.Generator(np.random.PCG64(seed))
    allowed = torch.zeros((R, S), dtype=torch.bool)
    bounds = np.linspace(0, R, n_samples + 1).astype(int)
    kb = np.linspace(0, S, n_samples + 1).astype(int)
    for b in range(n_samples):
        k0, k1 = int(kb[b]), int(kb[b + 1])
        runs = [(int(g.integers(k0, k1 - 40)), int(g.integers(8, 40))) for _ in range(max(4, (bounds[b + 1] - bounds[b]) // 2))]
        for r in range(bounds[b], bounds[b + 1]):
            for i in g.choice(len(runs), size=int(g.integers(1, 4)), replace=False):
                a, n_ = runs[i]
                allowed[r, a:a + n_] = True
    allowed[5] = False
    allowed[7] = False
    allowed[7, min(123, S - 1)] = True
    return allowed, torch.from_numpy(bounds.astype(np.int32))


@pytest.mark.parametrize('R,S,n_samples,lo,ordered', [(300, 4000, 1, True, True), (301, 6000, 3, True, True), (37, 700, 2, True, False),
                                                      (130, 40000, 2, True, True), (300, 4000, 1, False, True), (9, 300, 1, True, False)])
def test_xattn_group_tables_and_attention(dev, R, S, n_samples, lo, ordered):
    """Round 6, csrc/xattn_group.hip.  (1) mv2d_xattn_group_tables: groups = runs of 8 consecutive slots of every sample's order (never across samples;
    the rows behind the last sample form groups of their own), a group's list = the UNION of its members' CSR rows, every key once, mask bit j = member j
    lists it, sorted by (mask, key) inside 16384-key windows, padded to a multiple of 16 with mask-0 entries.  (2) mv2d_xattn_group_fwd == the per-query
    kernels (mv2d_xattn_fused_fwd) to fp32 rounding: the same products per (query, key) pair, the keys of a softmax row visited in union order."""
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    allowed, grp = _rect_pattern(R, S, 600 + R, n_samples)
    Rr = int(grp[-1])
    if R > 100:
        grp = grp.clone()
        grp[-1] = R - 3                                       # three bucket-padding rows behind the last sample
    row_ptr, col = O.csr_from_allowed(allowed)
    row_ptr, col, grp_d = row_ptr.to(dev), col.to(dev), grp.to(dev)
    order = None
    if ordered:
        order = torch.empty(R, dtype=torch.int32, device=dev)
        ops.xattn_query_order(row_ptr, col, grp_d, R, order, torch.zeros(2, dtype=torch.int32, device=dev))
    tab = ops.xattn_group_alloc(R, n_samples, int(col.numel()), dev)
    ops.xattn_group_tables(row_ptr, col, grp_d, R, tab, order=order)
    assert int(tab['ctl'][1]) == 0
    gt, ucol, umask = tab['g'].cpu().numpy(), tab['ucol'].cpu().numpy(), tab['umask'].cpu().numpy()
    ordr = order.cpu().numpy() if ordered else np.arange(R)
    rp, ci = row_ptr.cpu().numpy(), col.cpu().numpy()
    gl = grp.numpy().tolist() + [R]
    seen_rows, gi, used = [], 0, np.zeros(tab['ucap'], dtype=bool)
    for b in range(len(gl) - 1):
        for s0 in range(gl[b], gl[b + 1], 8):
            cnt = min(8, gl[b + 1] - s0)
            assert gt[0, gi] == s0 and gt[1, gi] == cnt, (gi, gt[:, gi], s0, cnt)
            members = ordr[s0:s0 + cnt]
            seen_rows += members.tolist()
            want = {}
            for j, r in enumerate(members):
                for k in ci[rp[r]:rp[r + 1]]:
                    want[int(k)] = want.get(int(k), 0) | (1 << j)
            n = int(gt[3, gi])
            assert n == len(want)
            if n:
                p0 = int(gt[2, gi])
                assert p0 % 16 == 0 and not used[p0:p0 + (n + 15) // 16 * 16].any()
                used[p0:p0 + (n + 15) // 16 * 16] = True
                keys, masks = ucol[p0:p0 + n], umask[p0:p0 + n]
                assert {int(k): int(m) for k, m in zip(keys, masks)} == want
                lo_k = min(want)
                sortkey = ((keys - lo_k) // 16384).astype(np.int64) * (1 << 40) + masks.astype(np.int64) * (1 << 32) + keys
                assert (np.diff(sortkey) > 0).all()
                pad = slice(p0 + n, p0 + (n + 15) // 16 * 16)
                assert (umask[pad] == 0).all() and ((ucol[pad] >= 0) & (ucol[pad] < S)).all()
            gi += 1
    assert sorted(seen_rows) == list(range(R)) and (gt[1, gi:] == 0).all()
    # ---- (2) the attention
    q = (rnd((R, 256), 661) * 0.3).to(dev)
    q[3] *= 8.0
    xk32, xv32 = rnd((S, 256), 662).to(dev), rnd((S, 256), 663).to(dev)
    Xk, Xk_lo = ops.f32_to_key16(xk32, with_lo=True)
    Xv, Xv_lo = ops.f32_to_key16(xv32, with_lo=True)
    if not lo:
        Xk_lo = Xv_lo = None
    Wk, Wv = rnd((256, 256), 664, 0.06).to(dev), rnd((256, 256), 666, 0.06).to(dev)
    bv = rnd((256,), 667).to(dev)
    WA, WB = ops.pack_xattn_maps(Wk, Wv)
    for empty_nan in (False, True):
        ref = ops.xattn_fused(q, WA, WB, bv, Xk, Xv, row_ptr, col, empty_nan=empty_nan, Xk_lo=Xk_lo, Xv_lo=Xv_lo)
        out = torch.full((R, 256), 7.0, device=dev)
        ops.xattn_group(q, WA, WB, bv, Xk, Xv, row_ptr, tab, out=out, empty_nan=empty_nan, Xk_lo=Xk_lo, Xv_lo=Xv_lo, order=order)
        assert torch.equal(torch.isnan(out), torch.isnan(ref))
        err = float((out - ref).abs().nan_to_num(0).max()) / float(ref.abs().nan_to_num(0).max())
        assert err < 2e-6, (empty_nan, err)
        out2 = torch.empty_like(out)
        ops.xattn_group(q, WA, WB, bv, Xk, Xv, row_ptr, tab, out=out2, empty_nan=empty_nan, Xk_lo=Xk_lo, Xv_lo=Xv_lo, order=order)
        assert torch.equal(out2.view(torch.int32), out.view(torch.int32))          # run-to-run bitwise
    assert bool(torch.isnan(out[5]).all()) and bool(torch.isfinite(out[6:]).all())
    print(f'xattn_group R {R} S {S}: rel err vs per-query kernels {err:.2e}; union {int(tab["ctl"][0])} entries for {int(col.numel())} pairs')


@pytest.mark.parametrize('R,S,dens,waves', [(37, 500, 0.05, 4), (37, 500, 0.05, 1), (301, 5000, 0.02, 8), (301, 5000, 0.02, 2), (64, 49 * 64, -1.0, 2),
                                            (2500, 3000, 0.01, 2), (1100, 49 * 1100, -1.0, 4),
                                            (20, 2000, 0.3, 4), (20, 2000, 0.3, 8), (20, 2000, 0.3, 1)])
def test_xattn_tile_equals_projected_attention(dev, R, S, dens, waves):
    """The default cross-attention route (csrc/xattn_tile.hip): query map -> MFMA tile attention on the UNPROJECTED rows -> context map
    == masked attention on K = Xk Wk^T + bk, V = Xv Wv^T + bv in fp64 (PETRMultiheadAttention's in_proj + core, MU/petr_transformer.py:487-513).
    Asymmetric random operands, rows of 1..600 keys (several tiles per wave, online-softmax rescales), an empty row, the RoI pattern of the
    S path; the intermediate operands are checked too, so that a transposed fragment cannot hide behind the softmax."""
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    g = np.random.Generator(np.random.PCG64(360 + R))
    if dens > 0:
        allowed = torch.from_numpy(g.random((R, S)) < dens)
        allowed[5] = False
        allowed[7, :] = False
        allowed[7, 123] = True
    else:
        allowed = torch.zeros((R, S), dtype=torch.bool)
        for r in range(R):
            allowed[r, r * 49:(r + 1) * 49] = True
            allowed[r, ((r + 3) % R) * 49:((r + 3) % R) * 49 + 49] = True
    q = (rnd((R, 256), 361) * 0.3).to(dev)
    q[3] *= 8.0                                                       # one query with sharp logits: the running maximum jumps between tiles
    Xk = rnd((S, 256), 362).to(dev).to(ops.key16_dtype())             # key16 rows (fp16 since round 4)
    Xv = rnd((S, 256), 363).to(dev).to(ops.key16_dtype())
    Wk = rnd((256, 256), 364, 0.06).to(dev); bk = rnd((256,), 365).to(dev)
    Wv = rnd((256, 256), 366, 0.06).to(dev); bv = rnd((256,), 367).to(dev)
    row_ptr, col = O.csr_from_allowed(allowed)
    row_ptr, col = row_ptr.to(dev), col.to(dev)
    ref = O.masked_cross_attention(q.double(), Xk.double() @ Wk.double().T + bk.double(), Xv.double() @ Wv.double().T + bv.double(), allowed.to(dev))
    WA, WB = ops.pack_xattn_maps(Wk, Wv)
    Qt = ops.xattn_qmap(q, WA)
    qk_ref = torch.einsum('rhd,hdc->rhc', q.double().view(R, 8, 32), Wk.double().view(8, 32, 256)).cpu()
    hi, lo = _unpack_qt(Qt, R)
    assert float((hi + lo - qk_ref).abs().max() / qk_ref.abs().max()) < 3e-5          # hi + lo = the fp32-class map
    assert float((hi - qk_ref).abs().max() / qk_ref.abs().max()) < (8e-4 if ops.key16_dtype() == torch.float16 else 6e-3)      # hi alone = its key16 rounding
    nnz = int(col.numel())
    dbg = torch.zeros((8, nnz), device=dev)
    z = ops.xattn_tile(Qt, Xk, Xv, row_ptr, col, empty_nan=False, waves=waves, dbg_logits=dbg)
    # logits (without the per-row constant q_h . bk_h) and z against fp64 on the same bf16 rows
    rows = torch.repeat_interleave(torch.arange(R), allowed.sum(1))
    lg_ref = torch.einsum('ehc,ec->he', qk_ref[rows], Xk.double().cpu()[col.cpu().long()])
    assert float((dbg.double().cpu() - lg_ref).abs().max() / lg_ref.abs().max()) < 2e-5
    att = torch.einsum('rhc,sc->hrs', qk_ref, Xk.double().cpu()).masked_fill(~allowed[None], float('-inf')).softmax(-1)
    att = torch.where(allowed.any(1)[None, :, None], att, torch.zeros_like(att))
    z_ref = torch.einsum('hrs,sc->rhc', att, Xv.double().cpu())
    assert relerr(z, z_ref) < 3e-5
    ctx = ops.xattn_ctxmap(z, WB, bv, row_ptr, empty_nan=False)
    has = allowed.any(1).to(dev)
    assert relerr(ctx[has], ref[has]) < 5e-5
    assert float(ctx[~has].abs().max()) == 0.0 if bool((~has).any()) else True          # 'zero' policy: no value bias for a query without keys
    z2 = ops.xattn_tile(Qt, Xk, Xv, row_ptr, col, empty_nan=False, waves=waves)
    assert torch.equal(z2, z)                                                           # deterministic, debug output does not change the result
    perm = torch.randperm(R, generator=torch.Generator().manual_seed(R)).to(torch.int32).to(dev)
    z4 = ops.xattn_tile(Qt, Xk, Xv, row_ptr, col, empty_nan=False, waves=waves, order=perm)       # any launch order of the blocks: bitwise the same rows
    assert torch.equal(z4, z)
    if dens > 0:
        assert float(z[5].abs().max()) == 0.0
        zn = ops.xattn_tile(Qt, Xk, Xv, row_ptr, col, waves=waves)
        assert bool(torch.isnan(zn[5]).all()) and torch.equal(zn[6:], z[6:])
        cn = ops.xattn_ctxmap(zn, WB, bv, row_ptr)
        assert bool(torch.isnan(cn[5]).all()) and torch.equal(cn[6:], ctx[6:])


# ------------------------------------------------------------------------------------------ geometry
def _problem(name):
    prob = synthetic.make_problem(name, seed=0)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    return prob, props


@pytest.mark.parametrize('name', ['micro_t', 'cfg1_t', 'cfg3_t'])
def test_box_params_roialign_refpoint(dev, name):
    from mv2d_amd import calib, ops
    from oracle import mv2d_oracle as O
    prob, props = _problem(name)
    metas = prob['img_metas']
    feat = torch.from_numpy(prob['feat'])
    V, C, h, w = feat.shape
    rois = O.bbox2roi(props)
    R = rois.shape[0]
    K_ref, E_ref = O.get_box_params(props, [m['intrinsics'] for m in metas], [m['extrinsics'] for m in metas])
    intr_ref = O.process_intrins_feat(rois, K_ref)
    ft = calib.frame_tables(metas, h, w)
    ct = calib.constant_tables()
    K_roi = torch.empty((R, 16), dtype=torch.float64, device=dev)
    intr = torch.zeros((R, 32), device=dev)
    minv = torch.empty((R, 16), device=dev)
    ops.box_params(rois.to(dev), ft['viewK'].to(dev), ft['viewE'].to(dev), intr, 32, minv, K_roi=K_roi)
    assert torch.equal(K_roi.cpu().view(R, 4, 4), K_ref)                              # fp64, same op order: bit-exact
    assert torch.equal(intr[:, :16].cpu(), intr_ref.clamp(-5e3, 5e3))
    minv_ref = torch.inverse(torch.bmm(K_ref, E_ref.transpose(1, 2))).float()
    assert relerr(minv.view(R, 4, 4), minv_ref) < 1e-6
    # center2lidar + normalise + posemb on oracle-provided center_pred
    cp = rnd((R, 3), 70).abs() * torch.tensor([3.0, 3.0, 20.0]) + torch.tensor([0.5, 0.5, 2.0])
    xyz_ref = O.center2lidar(cp, K_ref, E_ref)
    ref_ref = O.normalize_ref(xyz_ref)
    pos_ref = O.pos2posemb3d(ref_ref)
    xyz = torch.empty((R, 3), device=dev); ref = torch.empty((R, 3), device=dev); pos = torch.empty((R, 384), device=dev)
    ops.refpoint_posemb(cp.to(dev), 3, minv, ct['dim_t'].to(dev), xyz, ref, pos, R, torch.tensor(O.PC_RANGE, dtype=torch.float32))
    assert relerr(xyz, xyz_ref) < 1e-5
    assert relerr(ref, ref_ref) < 1e-5
    assert float((pos.cpu() - pos_ref).abs().max()) < 2e-5
    # RoIAlign (fp32 output) vs the oracle restatement
    if name != 'cfg3_t':
        fcl = ops.nchw_to_nhwc(feat.to(dev))
        out = torch.empty((R, 49, 256), device=dev)
        k16 = ops.key16_dtype()
        outb = torch.empty((R, 49, 256), device=dev, dtype=k16)
        outl = torch.empty((R, 49, 256), device=dev, dtype=k16)
        ops.roi_align(fcl, rois.to(dev), h, w, out0=outb, out0_f32=out, out0_lo=outl)
        ra_ref = O.roi_align(feat, rois).flatten(2).transpose(1, 2)
        assert relerr(out, ra_ref) < 1e-6
        assert torch.equal(outb, out.to(k16))                                        # the key16 output = the rounded fp32 output
        assert torch.equal(outl, (out - outb.float()).to(k16))                       # ... and its remainder (index-exact route: hi + lo rows)
        assert float((outb.double() + outl.double() - out.double()).abs().max()) < (1e-6 if k16 == torch.float16 else 1e-4) * float(out.abs().max())


@pytest.mark.parametrize('name,topk,expand', [('micro_t', 20, 2), ('cfg1_t', 20, 2), ('cfg1_s', 1, 0), ('cfg3_t', 20, 2), ('cfg2_s', 1, 0), ('nc6_s', 1, 0)])
def test_box_correlation_and_csr_bit_exact(dev, name, topk, expand):
    from mv2d_amd import calib, ops
    from oracle import mv2d_oracle as O
    prob, props = _problem(name)
    metas = prob['img_metas']
    V = len(metas)
    h, w = prob['feat'].shape[2:]
    rois = O.bbox2roi(props)
    R = rois.shape[0]
    npv = [len(p) for p in props]
    ft = calib.frame_tables(metas, h, w)
    ct = calib.constant_tables()
    trans_ref = O.view_transforms(metas)
    assert torch.equal(ft['trans'].view(V, V, 4, 4), trans_ref)
    ep = O.epipolar_in_box(rois, npv, metas[0]['pad_shape'], trans_ref, topk)
    vs = torch.tensor(np.concatenate([[0], np.cumsum(npv)]), dtype=torch.int32)
    match = torch.full((R, V, topk), -7, dtype=torch.int32, device=dev)
    ops.box_correlation(rois.to(dev), vs.to(dev), ft['trans'].to(dev), ct['lin'].to(dev), ct['depths'].to(dev), match, V, topk,
                        ft['pad_h'], ft['pad_w'], max(npv))
    m = match.cpu().view(R, -1)
    n_mismatch = 0
    for r in range(R):
        got = [int(x) for x in m[r] if x >= 0]
        exp = [i for (i, k) in ep[r] if k]
        n_mismatch += int(got != exp)
    assert n_mismatch == 0, f'{n_mismatch}/{R} RoIs with a different correlated-RoI list'
    if expand == 0:
        # S-path CSR over RoI-feature rows
        row_ptr = torch.empty(R + 1, dtype=torch.int32, device=dev)
        col = torch.empty(R * (1 + V * topk) * 49, dtype=torch.int32, device=dev)
        nnz = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.csr_from_corr(match, row_ptr, col, nnz, R, V, topk)
        corr, cmask = O.gen_box_roi_correlation(rois, npv, metas, topk)
        exp_cols = []
        for r in range(R):
            for j in range(corr.shape[1]):
                if cmask[r, j]:
                    exp_cols += [int(corr[r, j]) * 49 + c for c in range(49)]
        assert int(nnz) == len(exp_cols)
        assert col[:int(nnz)].cpu().tolist() == exp_cols
        return
    # T-path: compacted key list + CSR vs the oracle's boolean masks
    P = V * h * w
    ffr = O.gen_box_correlation(rois, npv, metas, h, w, 16, expand, topk)             # [R,V,h,w] bool
    pad = O.padding_mask(metas, h, w)
    assert torch.equal(ft['pad_mask'].bool().view(V, h, w), pad)
    roi_mask = torch.zeros(P, dtype=torch.uint8, device=dev)
    rect = torch.empty((R, 5), dtype=torch.int32, device=dev)
    pos2s = torch.empty(P, dtype=torch.int32, device=dev)
    s2pos = torch.empty(P, dtype=torch.int32, device=dev)
    S_out = torch.zeros(1, dtype=torch.int32, device=dev)
    bits = torch.empty(ops.csr_workspace_bytes(R, V, h, w) // 4, dtype=torch.int32, device=dev)
    row_count = torch.empty(R, dtype=torch.int32, device=dev)
    row_ptr = torch.empty(R + 1, dtype=torch.int32, device=dev)
    col = torch.empty(R * 2048, dtype=torch.int32, device=dev)
    nnz = torch.zeros(2, dtype=torch.int32, device=dev)
    ops.mask_compact(rois.to(dev), match, ft['pad_mask'].to(dev), roi_mask, rect, pos2s, s2pos, S_out, bits, row_count, row_ptr,
                     col, nnz, R, V, h, w, topk, 16.0, float(expand))
    assert torch.equal(roi_mask.cpu().bool().view(V, h, w), ffr.any(0))               # roi_mask of RH/mv2d_t_head.py:84
    keep = (ffr.any(0) & ~pad).view(-1)
    S = int(S_out)
    assert S == int(keep.sum())
    assert torch.equal(s2pos[:S].cpu().long(), keep.nonzero()[:, 0])
    allowed = (ffr & ~pad[None]).view(R, -1)[:, keep]                                 # [R,S]
    rp_ref, col_ref = O.csr_from_allowed(allowed)
    assert torch.equal(row_ptr.cpu(), rp_ref)
    assert int(nnz[0]) == int(rp_ref[-1]) and int(nnz[1]) == 0
    assert torch.equal(col[:int(nnz[0])].cpu(), col_ref)


@pytest.mark.parametrize('name', ['micro_t', 'cfg1_t'])
def test_pe_inputs(dev, name):
    from mv2d_amd import calib, ops
    from oracle import mv2d_oracle as O
    prob, props = _problem(name)
    metas = prob['img_metas']
    feat = torch.from_numpy(prob['feat'])
    V, C, h, w = feat.shape
    P = V * h * w
    ft = calib.frame_tables(metas, h, w)
    ct = calib.constant_tables()
    g = np.random.Generator(np.random.PCG64(80))
    sel = np.sort(g.choice(P, size=P // 3, replace=False)).astype(np.int32)
    S = len(sel)
    s2pos = torch.from_numpy(sel).to(dev)
    S_dev = torch.tensor([S], dtype=torch.int32, device=dev)
    fcl = ops.nchw_to_nhwc(feat.to(dev))
    k16 = ops.key16_dtype()
    ulp = 1e-3 if k16 == torch.float16 else 1e-2                                      # one key16 ulp, relative
    A1 = torch.zeros((P, 192), dtype=k16, device=dev)
    A2 = torch.zeros((P, 384), dtype=k16, device=dev)
    Xb = torch.zeros((P, 256), dtype=k16, device=dev)
    Xf = torch.zeros((P, 256), device=dev)
    ops.pe_inputs(s2pos, S_dev, P, fcl, ft['img2lidar'].to(dev), ft['coords_w'].to(dev), ft['coords_h'].to(dev),
                  ft['coords_d'].to(dev), ft['embeds'].to(dev), ct['dim_t'].to(dev), A1, A2, Xb, Xf, V, h, w, 64,
                  torch.tensor(O.POST_RANGE, dtype=torch.float64))
    x3 = O.pe_frustum_input(metas, h, w)                                              # [V,192,h,w] fp32
    sin = O.sine_pe3d(O.padding_mask(metas, h, w)[None])[0]                           # [V,384,h,w]
    x3 = x3.permute(0, 2, 3, 1).reshape(P, 192)[sel]
    sin = sin.permute(0, 2, 3, 1).reshape(P, 384)[sel]
    fr = feat.permute(0, 2, 3, 1).reshape(P, 256)[sel]
    assert torch.equal(Xf[:S].cpu(), fr)
    assert torch.equal(Xb[:S].cpu(), fr.to(k16))
    # key16-rounded outputs: compare with the key16 rounding of the oracle values (allow 1 ulp for ulp-level log / sin differences)
    d1 = (A1[:S].float().cpu() - x3.to(k16).float()).abs()
    assert float((d1 / x3.abs().clamp_min(1.0)).max()) < ulp
    assert float((d1 > 0).float().mean()) < (1e-2 if k16 == torch.float16 else 1e-3)
    d2 = (A2[:S].float().cpu() - sin.to(k16).float()).abs()
    assert float(d2.max()) < ulp
    assert float((d2 > 0).float().mean()) < (1e-1 if k16 == torch.float16 else 2e-2)
    assert float(A1[S:].float().abs().max()) == 0.0


@pytest.mark.parametrize('name', ['cfg1_t', 'cfg1_s'])
def test_pe_frustum_rows_fast_equals_reference_order(dev, name):
    """mv2d_pe_frustum_f32 (round 5: linear-in-depth point, reciprocal ranges, table-driven fp64 log x1 - log x2) against (a) the oracle's
    restatement of MU/pe.py:96-131 (fp64, .float()) and (b) pe_inputs_kernel<true> (the reference's operation order in fp64 on the device):
    the fp32 rows are EQUAL except where the fp64 value sits within ~1e-15 of a rounding boundary (none expected in 1e5 values; bound: 1 ulp
    in fewer than 1 of 1e4 elements)."""
    from mv2d_amd import calib, ops
    from oracle import mv2d_oracle as O
    prob, props = _problem(name)
    metas = prob['img_metas']
    feat = torch.from_numpy(prob['feat'])
    V, C, h, w = feat.shape
    P = V * h * w
    ft = calib.frame_tables(metas, h, w)
    ct = calib.constant_tables()
    g = np.random.Generator(np.random.PCG64(81))
    sel = np.sort(g.choice(P, size=P // 2, replace=False)).astype(np.int32)
    S = len(sel)
    s2pos = torch.from_numpy(sel).to(dev)
    S_dev = torch.tensor([S], dtype=torch.int32, device=dev)
    pr = torch.tensor(O.POST_RANGE, dtype=torch.float64)
    T = {k: ft[k].to(dev) for k in ('img2lidar', 'coords_w', 'coords_h', 'coords_d', 'embeds')}
    fast = torch.zeros((P, 192), device=dev)
    ops.pe_frustum_f32(s2pos, S_dev, P, T['img2lidar'], T['coords_w'], T['coords_h'], T['coords_d'], fast, V, h, w, 64, pr)
    k16 = ops.key16_dtype()
    slow = torch.zeros((P, 192), device=dev)
    ops.pe_inputs(s2pos, S_dev, P, ops.nchw_to_nhwc(feat.to(dev)), T['img2lidar'], T['coords_w'], T['coords_h'], T['coords_d'], T['embeds'],
                  ct['dim_t'].to(dev), torch.zeros((P, 192), dtype=k16, device=dev), torch.zeros((P, 384), dtype=k16, device=dev),
                  torch.zeros((P, 256), dtype=k16, device=dev), None, V, h, w, 64, pr, A_frustum_f32=slow, A_sine_f32=torch.zeros((P, 384), device=dev))
    ref = O.pe_frustum_input(metas, h, w).permute(0, 2, 3, 1).reshape(P, 192)[sel]
    for other, label in ((slow[:S].cpu(), 'pe_inputs<true>'), (ref, 'oracle')):
        d = (fast[:S].cpu() - other).abs()
        n_diff = int((d > 0).sum())
        print(f'[pe_frustum_f32 vs {label}] {name}: {n_diff} of {d.numel()} elements differ, max |diff| {float(d.max()):.2e}')
        assert n_diff <= max(1, d.numel() // 10000)
        assert float((d / other.abs().clamp_min(1e-3)).max()) < 3e-7
    assert float(fast[S:].abs().max()) == 0.0


def test_roi_tap_positions_cover_every_roialign_tap_exactly(dev):
    """mv2d_roi_positions with expand_stride < 0 (round 5): the marked rectangle of a RoI is exactly the bounding rectangle of the cells its
    RoIAlign taps touch (same fp32 sample coordinates as roi_align_kernel: aligned, adaptive grid) -- RoIs inside the image, across its
    borders, entirely outside, tiny and huge; so the S path's PE list holds every tap (RoIAlign(pe) is unchanged) and nothing else."""
    from mv2d_amd import ops
    H, W, V = 32, 88, 2
    g = np.random.Generator(np.random.PCG64(77))
    n = 400
    x0 = g.uniform(-60, 1408 + 40, n).astype(np.float32); y0 = g.uniform(-60, 512 + 40, n).astype(np.float32)
    bw = np.where(g.random(n) < 0.2, g.uniform(0.5, 8, n), g.uniform(8, 400, n)).astype(np.float32)
    bh = np.where(g.random(n) < 0.2, g.uniform(0.5, 8, n), g.uniform(8, 300, n)).astype(np.float32)
    rois = np.stack([g.integers(0, V, n).astype(np.float32), x0, y0, x0 + bw, y0 + bh], 1).astype(np.float32)
    rois[0] = [0, 0, 0, 1408, 512]                                   # the whole image
    rois[1] = [1, -500, -500, -400, -400]                            # entirely outside: no tap
    rois[2] = [0, 1400, 500, 1500, 600]                              # across the lower right corner
    P = V * H * W
    rt = torch.from_numpy(rois).to(dev)
    roi_mask = torch.zeros(P, dtype=torch.uint8, device=dev)
    rect = torch.empty((n, 5), dtype=torch.int32, device=dev)
    pos2s = torch.empty(P, dtype=torch.int32, device=dev); s2pos = torch.empty(P, dtype=torch.int32, device=dev)
    S = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.roi_positions(rt, torch.zeros(P, dtype=torch.uint8, device=dev), roi_mask, rect, pos2s, s2pos, S, n, V, H, W, stride=16.0, expand_stride=-1.0)
    rect = rect.cpu().numpy()
    f32 = np.float32
    exp_mask = np.zeros((V, H, W), bool)
    for r in range(n):
        v = int(rois[r, 0])
        cells = []
        for lo, hi, N in ((rois[r, 1], rois[r, 3], W), (rois[r, 2], rois[r, 4], H)):
            a, b = f32(lo * f32(0.0625) - f32(0.5)), f32(hi * f32(0.0625) - f32(0.5))
            ln = f32(b - a); bn = f32(ln / f32(7.0)); gr = int(np.ceil(f32(ln / f32(7.0))))
            touched = set()
            for pw in range(7):
                for ix in range(gr):
                    xx = f32(f32(a + f32(f32(pw) * bn)) + f32(f32(f32(ix + 0.5) * bn) / f32(gr)))
                    if xx < -1.0 or xx > N:
                        continue
                    x = max(float(xx), 0.0)
                    xl = int(x)
                    if xl >= N - 1:
                        xl = xh = N - 1
                    else:
                        xh = xl + 1
                    touched.update((xl, xh))
            cells.append(touched)
        tx, ty = cells
        if tx and ty:
            assert (rect[r, 1], rect[r, 2], rect[r, 3], rect[r, 4]) == (min(ty), max(ty), min(tx), max(tx)), (r, rois[r], rect[r])
            exp_mask[v, min(ty):max(ty) + 1, min(tx):max(tx) + 1] = True
        else:
            assert rect[r, 2] < rect[r, 1] or rect[r, 4] < rect[r, 3], (r, rois[r], rect[r])
    np.testing.assert_array_equal(roi_mask.cpu().numpy().astype(bool).reshape(V, H, W), exp_mask)
    assert int(S.item()) == int(exp_mask.sum())


def test_decode_topk_bit_exact(dev):
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    for R, seed in [(12, 90), (300, 91), (900, 92)]:
        cls = rnd((R, 10), seed, 2.0) - 3.0
        reg = rnd((R, 10), seed + 100)
        reg[:, 0] *= 40.0; reg[:, 1] *= 40.0; reg[:, 4] *= 6.0                         # some centres fall outside the range
        boxes = torch.zeros((300, 9), device=dev); scores = torch.zeros(300, device=dev)
        labels = torch.zeros(300, dtype=torch.int64, device=dev); bidx = torch.zeros(300, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.decode_topk(cls.to(dev), reg.to(dev), R, 10, 300, torch.tensor(O.POST_RANGE, dtype=torch.float32), boxes, scores,
                        labels, bidx, cnt)
        b_ref, s_ref, l_ref, i_ref = O.decode(cls, reg)
        n = int(cnt)
        assert n == b_ref.shape[0]
        assert torch.equal(labels[:n].cpu(), l_ref)                                   # bit-exact integers
        assert torch.equal(bidx[:n].cpu(), i_ref)
        assert relerr(scores[:n], s_ref) < 1e-6
        assert relerr(boxes[:n], b_ref) < 1e-5


def test_decode_topk_tie_order(dev):
    """torch.topk leaves the order of equal scores unspecified (CB/coders/nms_free_coder.py:66); the kernel
    defines it: equal logits are emitted lower flat index first."""
    from mv2d_amd import ops
    from oracle import mv2d_oracle as O
    R = 40
    cls = rnd((R, 10), 95, 2.0) - 3.0
    cls[3, 4] = cls[2, 1] = cls[30, 9] = 5.0                                          # three-way tie at the top
    reg = rnd((R, 10), 96) * 0.1
    boxes = torch.zeros((300, 9), device=dev); scores = torch.zeros(300, device=dev)
    labels = torch.zeros(300, dtype=torch.int64, device=dev); bidx = torch.zeros(300, dtype=torch.int64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.decode_topk(cls.to(dev), reg.to(dev), R, 10, 300, torch.tensor(O.POST_RANGE, dtype=torch.float32), boxes, scores, labels,
                    bidx, cnt)
    assert int(cnt) == 300
    assert bidx[:3].cpu().tolist() == [2, 3, 30] and labels[:3].cpu().tolist() == [1, 4, 9]
    assert bool((scores[:299] >= scores[1:300]).all())


def test_qg_conv_pool_fused(dev):
    """conv3x3 + ReLU + AvgPool2d(7) per RoI in one kernel == F.conv2d in fp64 on the key16-rounded operands; the split-precision variant
    (hi + lo cells and weights, index-exact route) == the same on the UNROUNDED operands."""
    from mv2d_amd import ops
    k16 = ops.key16_dtype()
    for R in (1, 37, 300, 1301):
        x = rnd((R, 256, 7, 7), 70).to(dev)
        w = rnd((256, 256, 3, 3), 71, 0.03).to(dev)
        b = rnd((256,), 72).to(dev)
        x32 = x.permute(0, 2, 3, 1).reshape(R, 49, 256).contiguous()
        w32 = w.permute(0, 2, 3, 1).reshape(256, 2304).contiguous()
        xr, wr = x32.to(k16), w32.to(k16)
        out = torch.empty((R, 256), device=dev)
        ops.qg_conv_pool(xr, ops.pack_key16(w32), b, out)
        xh, xl = ops.f32_to_key16(x32, with_lo=True)
        assert torch.equal(xh, xr)
        outx = torch.empty((R, 256), device=dev)
        ops.qg_conv_pool_x3(xh, xl, ops.pack_key16_x3(w32), b, outx)
        refx = F.avg_pool2d(F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)), 7).flatten(1)
        assert relerr(outx, refx) < (3e-6 if k16 == torch.float16 else 3e-5)
        ref = F.avg_pool2d(F.relu(F.conv2d(xr.float().view(R, 7, 7, 256).permute(0, 3, 1, 2).double(),
                                           wr.float().view(256, 3, 3, 256).permute(0, 3, 1, 2).double(), b.double(), padding=1)), 7).flatten(1)
        assert relerr(out, ref) < 1e-5


@pytest.mark.parametrize('M,use_mdev,use_ri', [(95, False, False), (97, False, True), (1000, True, True), (8794, True, False), (70349, False, True)])
def test_pe_fused_tab_kernel(dev, M, use_mdev, use_ri):
    """mv2d_pe_fused_tab (csrc/pe_tab96.hip: position_encoder MLP + SE gate + table row + feature row in one launch, key16 = fp16 operands and
    hidden layer): against the same chain in fp64 on the key16-rounded operands (hidden layer rounded to key16 like the kernel does), and the
    96-row / 8-wave shape (default) == the two-64-row-blocks-per-CU shape bit for bit."""
    from mv2d_amd import ops
    k16 = ops.key16_dtype()
    NP = M + 50 if use_ri else M
    A1 = rnd((M, 192), 90).to(dev).to(k16)
    Xf32 = rnd((NP, 256), 92).to(dev)
    ri = torch.randperm(NP, generator=torch.Generator().manual_seed(5))[:M].to(torch.int32).to(dev) if use_ri else None
    Xrows = Xf32[ri.long()] if use_ri else Xf32
    Xfb = Xrows.to(k16)
    W = {k: v.to(dev) for k, v in dict(w1a=rnd((1024, 192), 93, 0.08), w1b=rnd((256, 1024), 94, 0.04), wr=rnd((256, 256), 97, 0.07),
                                        we=rnd((256, 256), 98, 0.07)).items()}
    wp = {k: ops.pack_key16(v) for k, v in W.items()}
    bias = {k: rnd((n,), 99 + i).to(dev) for i, (k, n) in enumerate(dict(b1a=1024, b1b=256, br=256, be=256).items())}
    wp.update(bias)
    period = 37
    tab = rnd((period, 256), 131).to(dev)
    md = torch.tensor([M - 13], dtype=torch.int32, device=dev) if use_mdev else None
    outs = {}
    for shape in (1, 0):
        pe = torch.zeros((M, 256), device=dev); xk = torch.zeros((M, 256), device=dev, dtype=k16)
        ops.pe_fused_tab(A1, Xfb, Xf32, md, wp, tab, period, pe, xk, M=M, row_index=ri, shape=shape)
        torch.cuda.synchronize()
        outs[shape] = (pe, xk)
    Mv = M - 13 if use_mdev else M
    for shape in (1, 0):                                         # Xk is optional (S path): pe alone is the same
        pe_only = torch.zeros((M, 256), device=dev)
        ops.pe_fused_tab(A1, Xfb, Xf32, md, wp, tab, period, pe_only, None, M=M, row_index=ri, shape=shape)
        assert torch.equal(pe_only[:Mv], outs[1][0][:Mv])
    assert torch.equal(outs[1][0][:Mv], outs[0][0][:Mv])
    assert torch.equal(outs[1][1][:Mv].view(torch.int16), outs[0][1][:Mv].view(torch.int16))
    assert not outs[1][0][Mv:].any() and outs[1][0][:Mv].abs().sum() > 0
    # fp64 reference on the rounded operands
    r = lambda t: t.to(k16).double()                                                   # noqa: E731
    h1 = torch.relu(A1.double() @ r(W['w1a']).T + bias['b1a'].double()).float().to(k16).double()
    p1 = h1 @ r(W['w1b']).T + bias['b1b'].double()
    hg = torch.relu(Xfb.double() @ r(W['wr']).T + bias['br'].double()).float().to(k16).double()
    gate = torch.sigmoid(hg @ r(W['we']).T + bias['be'].double())
    pos = (ri.long() if use_ri else torch.arange(M, device=dev)) % period
    pe_ref = tab.double()[pos] + p1 * gate
    assert relerr(outs[1][0][:Mv], pe_ref[:Mv]) < 1e-4                                  # (a hidden value on a rounding boundary may flip one key16 ulp)
    xk_ref = pe_ref + Xrows.double()
    assert relerr(outs[1][1][:Mv].double(), xk_ref[:Mv]) < (1e-3 if k16 == torch.float16 else 8e-3)


@pytest.mark.parametrize('M,use_mdev,use_ri,rows', [(63, False, False, False), (130, False, True, True), (1000, True, True, True), (8794, True, False, False)])
def test_pe_fused_x3_kernel(dev, M, use_mdev, use_ri, rows):
    """mv2d_pe_fused_x3 (csrc/pe_x3.hip, index-exact route): the PE block in split precision on UNROUNDED fp32 inputs against fp64 -- pe at
    fp32-class accuracy; with `rows` the key rows pe + feat and the value rows feat come out as key16 hi + lo pairs."""
    from mv2d_amd import ops
    k16 = ops.key16_dtype()
    NP = M + 50 if use_ri else M
    A1 = (rnd((M, 192), 190) * 3.0).to(dev)
    Xmap = rnd((NP, 256), 192).to(dev)
    ri = torch.randperm(NP, generator=torch.Generator().manual_seed(6))[:M].to(torch.int32).to(dev) if use_ri else None
    Xrows = Xmap[ri.long()] if use_ri else Xmap
    W = {k: v.to(dev) for k, v in dict(w1a=rnd((1024, 192), 193, 0.08), w1b=rnd((256, 1024), 194, 0.04), wr=rnd((256, 256), 197, 0.07),
                                        we=rnd((256, 256), 198, 0.07)).items()}
    bias = {k: rnd((n,), 199 + i).to(dev) for i, (k, n) in enumerate(dict(b1a=1024, b1b=256, br=256, be=256).items())}
    wx = {k: ops.pack_x3(v) for k, v in W.items()}
    wx.update(bias)
    period = 41
    tab = rnd((period, 256), 231).to(dev)
    md = torch.tensor([M - 13], dtype=torch.int32, device=dev) if use_mdev else None
    pe = torch.zeros((M, 256), device=dev)
    pairs = [tuple(torch.zeros((M, 256), device=dev, dtype=k16) for _ in range(2)) for _ in range(2)] if rows else [None, None]
    ops.pe_fused_x3(A1, Xmap, md, wx, tab, period, pe=pe, Xk=pairs[0], Xv=pairs[1], M=M, row_index=ri)
    torch.cuda.synchronize()
    Mv = M - 13 if use_mdev else M
    d = lambda t: t.double()                                                            # noqa: E731
    p1 = torch.relu(d(A1) @ d(W['w1a']).T + d(bias['b1a'])) @ d(W['w1b']).T + d(bias['b1b'])
    gate = torch.sigmoid(torch.relu(d(Xrows) @ d(W['wr']).T + d(bias['br'])) @ d(W['we']).T + d(bias['be']))
    pos = (ri.long() if use_ri else torch.arange(M, device=dev)) % period
    pe_ref = d(tab)[pos] + p1 * gate
    assert relerr(pe[:Mv], pe_ref[:Mv]) < 2e-5                                          # bf16x3: 2^-17 per operand
    assert not pe[Mv:].any()
    if rows:
        tol = 2e-6 if k16 == torch.float16 else 3e-5
        assert relerr(d(pairs[0][0][:Mv]) + d(pairs[0][1][:Mv]), (pe_ref + d(Xrows))[:Mv]) < 2e-5 + tol
        assert relerr(d(pairs[1][0][:Mv]) + d(pairs[1][1][:Mv]), d(Xrows)[:Mv]) < tol
        assert torch.equal(pairs[1][0][:Mv], Xrows[:Mv].to(k16))
        pe_only = torch.zeros((M, 256), device=dev)
        ops.pe_fused_x3(A1, Xmap, md, wx, tab, period, pe=pe_only, M=M, row_index=ri)
        assert torch.equal(pe_only, pe)
    # round 6: the second shape of the kernel (csrc/pe_x3b.hip: rows owned by waves, hidden layer in registers, weights through an LDS ring) does the
    # same products in the same k order: BITWISE the same outputs
    wx['w1a_p'], wx['wr_p'] = ops.pack_x3_rowperm(W['w1a']), ops.pack_x3_rowperm(W['wr'])
    pe_b = torch.zeros((M, 256), device=dev)
    pairs_b = [tuple(torch.zeros((M, 256), device=dev, dtype=k16) for _ in range(2)) for _ in range(2)] if rows else [None, None]
    ops.pe_fused_x3b(A1, Xmap, md, wx, tab, period, pe=pe_b, Xk=pairs_b[0], Xv=pairs_b[1], M=M, row_index=ri)
    assert torch.equal(pe_b.view(torch.int32), pe.view(torch.int32)), float((pe_b - pe).abs().max())
    if rows:
        for a, b in zip(pairs_b, pairs):
            assert torch.equal(a[0].view(torch.int16), b[0].view(torch.int16)) and torch.equal(a[1].view(torch.int16), b[1].view(torch.int16))
        nop = [tuple(torch.zeros((M, 256), device=dev, dtype=k16) for _ in range(2)) for _ in range(2)]
        ops.pe_fused_x3b(A1, Xmap, md, wx, tab, period, Xk=nop[0], Xv=nop[1], M=M, row_index=ri)          # (T path: no pe output)
        assert torch.equal(nop[0][0].view(torch.int16), pairs[0][0].view(torch.int16))
        # round 6: the pe rows written at row row_index[m] of a position-indexed map (what RoIAlign then reads without the position -> row table)
        if use_ri:
            for fn in (ops.pe_fused_x3, ops.pe_fused_x3b):
                pe_map = torch.full((NP, 256), 7.0, device=dev)
                fn(A1, Xmap, md, wx, tab, period, pe=pe_map, M=M, row_index=ri, pe_at_index=True)
                assert torch.equal(pe_map[ri.long()[:Mv]].view(torch.int32), pe[:Mv].view(torch.int32))
                rest = torch.ones(NP, dtype=torch.bool, device=dev)
                rest[ri.long()[:Mv]] = False
                assert bool((pe_map[rest] == 7.0).all())
        if k16 == torch.float16:
            # round 6: the lo halves as e4m3 "lo8" rows (256 B): the bytes are the encoding of the key16 lo halves, the hi rows are untouched; both kernels
            for fn in (ops.pe_fused_x3, ops.pe_fused_x3b):
                p8 = [(torch.zeros((M, 256), device=dev, dtype=k16), torch.zeros((M, 256), device=dev, dtype=torch.uint8)) for _ in range(2)]
                fn(A1, Xmap, md, wx, tab, period, Xk=p8[0], Xv=p8[1], M=M, row_index=ri)
                for a, b in zip(p8, pairs):
                    assert torch.equal(a[0].view(torch.int16), b[0].view(torch.int16))
                    assert torch.equal(a[1][:Mv], ops.lo8_encode(b[1][:Mv])) and not a[1][Mv:].any()


def test_attn_out_fused_x3(dev):
    """split-precision (bf16x3) row-fused out_proj + LN (+ q proj): fp32-class accuracy against fp64."""
    from mv2d_amd import ops
    for M in (300, 37):
        ctx, res, qpos = rnd((M, 256), 110).to(dev), rnd((M, 256), 111).to(dev), rnd((M, 256), 112).to(dev)
        Wo, Wq = rnd((256, 256), 113, 0.06).to(dev), rnd((256, 256), 114, 0.06).to(dev)
        bo, bq, lw, lb = rnd((256,), 115).to(dev), rnd((256,), 116).to(dev), rnd((256,), 117).to(dev), rnd((256,), 118).to(dev)
        x1 = torch.empty((M, 256), device=dev); q = torch.empty((M, 256), device=dev)
        ops.attn_out_fused_x3(ctx, res, ops.pack_x3(Wo), bo, (lw, lb), x1, qpos=qpos, Wq_x3=ops.pack_x3(Wq), bq=bq, qscale=0.25, q_out=q)
        xr = F.layer_norm(ctx.double() @ Wo.double().T + bo.double() + res.double(), (256,), lw.double(), lb.double())
        qr = ((xr + qpos.double()) @ Wq.double().T + bq.double()) * 0.25
        assert relerr(x1, xr) < 3e-5 and relerr(q, qr) < 3e-5
        x2 = torch.empty((M, 256), device=dev)
        ops.attn_out_fused_x3(ctx, res, ops.pack_x3(Wo), bo, (lw, lb), x2)
        assert torch.equal(x1, x2)


def test_ffn_out_fused_x3(dev):
    """FFN tail (slab sum + b2 + residual + LN + post_norm) + next layer's in_proj (bf16x3) in one kernel, against fp64."""
    from mv2d_amd import ops
    for M in (300, 37):
        parts = rnd((32, M, 256), 120, 0.3).to(dev)
        b2, res, qpos = rnd((256,), 121).to(dev), rnd((M, 256), 122).to(dev), rnd((M, 256), 123).to(dev)
        lw, lb, pw, pb = rnd((256,), 124).to(dev), rnd((256,), 125).to(dev), rnd((256,), 126).to(dev), rnd((256,), 127).to(dev)
        Win, b_in = rnd((768, 256), 128, 0.06).to(dev), rnd((768,), 129).to(dev)
        x = torch.empty((M, 256), device=dev); xq = torch.empty_like(x); outs = torch.empty_like(x); qkv = torch.empty((M, 768), device=dev)
        ops.ffn_out_fused_x3(parts, b2, res, (lw, lb), (pw, pb), x, qpos, xq, outs=outs, Win_x3=ops.pack_x3(Win), b_in=b_in, qkv=qkv)
        y = F.layer_norm(parts.double().sum(0) + b2.double() + res.double(), (256,), lw.double(), lb.double())
        assert relerr(x, y) < 1e-5 and relerr(xq, y + qpos.double()) < 1e-5
        assert relerr(outs, F.layer_norm(y, (256,), pw.double(), pb.double())) < 1e-5
        yq = y + qpos.double()
        ref = torch.cat([yq @ Win[:512].double().T, y @ Win[512:].double().T], 1) + b_in.double()
        assert relerr(qkv, ref) < 3e-5
        # same LN as row_ln (bit-identical), and the last-layer form without in_proj
        x_r = torch.empty_like(x); xq_r = torch.empty_like(x); o_r = torch.empty_like(x)
        ops.row_ln(parts, bias=b2, residual=res, ln=(lw, lb), out=x_r, addvec=qpos, out_plus=xq_r, ln2=(pw, pb), out2=o_r)
        assert torch.equal(x, x_r) and torch.equal(xq, xq_r) and torch.equal(outs, o_r)
        x2 = torch.empty_like(x); xq2 = torch.empty_like(x)
        ops.ffn_out_fused_x3(parts, b2, res, (lw, lb), None, x2, qpos, xq2)
        assert torch.equal(x2, x)


@pytest.mark.parametrize('R', [300, 37, 900])
def test_sa_block_fused_x3(dev, R):
    """self-attention core + out_proj + LN + q projection in one kernel == self_attn followed by attn_out_fused_x3 (same
    arithmetic after the context tile; the attention partial states are merged from 2 instead of 4 waves per head)."""
    from mv2d_amd import ops
    qkv = rnd((R, 768), 130).to(dev)
    res, qpos = rnd((R, 256), 131).to(dev), rnd((R, 256), 132).to(dev)
    Wo, Wq = ops.pack_x3(rnd((256, 256), 133, 0.06).to(dev)), ops.pack_x3(rnd((256, 256), 134, 0.06).to(dev))
    bo, bq, lw, lb = rnd((256,), 135).to(dev), rnd((256,), 136).to(dev), rnd((256,), 137).to(dev), rnd((256,), 138).to(dev)
    ctx = torch.empty((R, 256), device=dev)
    ops.self_attn(qkv, ctx, R)
    x_ref = torch.empty((R, 256), device=dev); q_ref = torch.empty((R, 256), device=dev)
    ops.attn_out_fused_x3(ctx, res, Wo, bo, (lw, lb), x_ref, qpos=qpos, Wq_x3=Wq, bq=bq, qscale=0.25, q_out=q_ref)
    x1 = torch.empty((R, 256), device=dev); q1 = torch.empty((R, 256), device=dev)
    ops.sa_block_fused_x3(qkv, res, Wo, bo, (lw, lb), x1, qpos=qpos, Wq_x3=Wq, bq=bq, qscale=0.25, q_out=q1)
    assert relerr(x1, x_ref) < 2e-5 and relerr(q1, q_ref) < 2e-5


def test_query_embed_fused_x3(dev):
    """fc_center + center2lidar + normalisation + pos2posemb3d + query_embedding (bf16x3) in one kernel == the four-launch chain."""
    from mv2d_amd import ops, calib
    for R in (300, 37):
        enc2 = rnd((R, 256), 140).to(dev)
        Wc = rnd((3, 256), 141, 0.05).to(dev); bc = torch.tensor([3.5, 3.5, 25.0], device=dev)
        minv = rnd((R, 16), 142, 0.5).to(dev)
        W0, b0, W2, b2 = rnd((256, 384), 143, 0.06).to(dev), rnd((256,), 144).to(dev), rnd((256, 256), 145, 0.06).to(dev), rnd((256,), 146).to(dev)
        dim_t = calib.constant_tables()['dim_t'].to(dev)
        pc = torch.tensor([-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])
        center = ops.gemm_f32(enc2, Wc, bc)
        xyz, ref, posemb = torch.empty((R, 3), device=dev), torch.empty((R, 3), device=dev), torch.empty((R, 384), device=dev)
        ops.refpoint_posemb(center, 3, minv, dim_t, xyz, ref, posemb, R, pc)
        qpos_ref = ops.gemm_f32(ops.gemm_f32(posemb, W0, b0, act=1), W2, b2)
        c2, x2, r2, p2, q2 = (torch.empty((R, 3), device=dev), torch.empty((R, 3), device=dev), torch.empty((R, 3), device=dev),
                              torch.empty((R, 384), device=dev), torch.empty((R, 256), device=dev))
        ops.query_embed_fused_x3(enc2, Wc, bc, minv, dim_t, pc, ops.pack_x3(W0), b0, ops.pack_x3(W2), b2, c2, x2, r2, p2, q2)
        assert relerr(c2, center) < 1e-5 and relerr(x2, xyz) < 1e-4 and relerr(r2, ref) < 1e-4
        assert float((p2 - posemb).abs().max()) < 5e-3          # sin/cos of arguments up to ~2 pi * ref / 1: differences of the fp32 centre propagate
        assert relerr(q2, qpos_ref) < 5e-3
