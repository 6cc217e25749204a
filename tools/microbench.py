#!/usr/bin/env python
"""Per-kernel micro-benchmarks on the GPU box (HIP events, back-to-back launches on one stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import ops

dev = torch.device('cuda:0')
BF = torch.bfloat16


def timeit(fn, reps=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def bench_bf16(M, N, K, out_f32=False, name=''):
    A = torch.randn(M, K, device=dev).to(BF)
    W = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    b = torch.randn(N, device=dev)
    out = torch.empty((M, N), device=dev, dtype=torch.float32 if out_f32 else BF)
    us = timeit(lambda: ops.gemm_bf16(A, W, b, out=out))
    fl = 2.0 * M * N * K
    by = (M * K + N * K) * 2 + M * N * (4 if out_f32 else 2)
    print(f'gemm_bf16 {name:12s} M={M:6d} N={N:5d} K={K:5d} out={"f32" if out_f32 else "bf16"}: {us:8.2f} us  {fl / us / 1e6:8.1f} TF/s  {by / us / 1e3:8.1f} GB/s')


def bench_f32(M, N, K, split=1):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    out = torch.empty((split, M, N) if split > 1 else (M, N), device=dev)
    us = timeit(lambda: ops.gemm_f32(A, W, b, split_k=split, out=out))
    print(f'gemm_f32  M={M:5d} N={N:5d} K={K:5d} split={split}: {us:8.2f} us  {2.0 * M * N * K / us / 1e6:7.2f} TF/s')


if __name__ == '__main__':
    x = torch.randn(8, 256, device=dev)
    w = torch.ones(256, device=dev)
    print(f'row_ln 8 rows (launch floor): {timeit(lambda: ops.row_ln(x, ln=(w, w))):.2f} us')
    x = torch.randn(300, 256, device=dev)
    print(f'row_ln 300 rows: {timeit(lambda: ops.row_ln(x, ln=(w, w))):.2f} us')
    for K in (64, 256, 1024):
        bench_bf16(8832, 256, K, name='pe_out')
    bench_bf16(8832, 256, 1024, out_f32=True, name='pe_out_f32')
    bench_bf16(8832, 1024, 192, name='pe_in1')
    bench_bf16(8832, 1024, 384, name='pe_in2')
    bench_bf16(14700, 3072, 256, name='kv')
    bench_bf16(16384, 4096, 4096, name='big')
    bench_bf16(4096, 4096, 4096, name='big4k')
    bench_f32(300, 256, 256)
    bench_f32(300, 768, 256)
    bench_f32(300, 2048, 256)
    bench_f32(300, 256, 2048, split=8)
    bench_f32(900, 256, 256)
    qkv = torch.randn(300, 768, device=dev)
    print(f'self_attn R=300: {timeit(lambda: ops.self_attn(qkv)):.2f} us')
    qkv = torch.randn(900, 768, device=dev)
    print(f'self_attn R=900: {timeit(lambda: ops.self_attn(qkv)):.2f} us')
