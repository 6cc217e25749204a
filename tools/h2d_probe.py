import torch, time
dev = torch.device('cuda:0')
big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
x = torch.randn(4096, 4096, device=dev)
for size in (128 << 10, 256 << 10, 320 << 10, 384 << 10, 400 << 10, 448 << 10, 512 << 10, 1 << 20, 4 << 20):
    h = torch.empty(size, dtype=torch.uint8).pin_memory()
    d = torch.empty(size, dtype=torch.uint8, device=dev)
    for view in (False, True):
        hs = h[: size - 64] if view else h
        ds = d[: size - 64] if view else d
        ds.copy_(hs, non_blocking=True); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            for _ in range(3): y = x @ x          # GPU busy for a few ms
            t0 = time.perf_counter(); ds.copy_(hs, non_blocking=True); ts.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        print(size >> 10, 'KB', 'view' if view else 'whole', 'pinned' if hs.is_pinned() else 'NOT PINNED', 'call %.3f ms' % (sorted(ts)[2] * 1e3))
