"""Practical HBM write / copy bandwidth on this GPU (graph-replayed torch fill_/copy_, HIP events)."""
import torch
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for mb in (90, 360, 1440):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
    t = timeit(lambda: a.fill_(1.0)); print(f'fill {mb} MB: {t:.1f} us  {mb*1.048576/t*1e0:.2f} TB/s'.replace('TB/s','TB/s'))
    t = timeit(lambda: b.copy_(a)); print(f'copy {mb} MB: {t:.1f} us  {2*mb*1.048576/t:.2f} TB/s (read+write)')
