"""Which HIP streams really run concurrently?  Pairwise overlap matrix of N streams (spin-kernel chains)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [torch.cuda.Stream(device=dev) for _ in range(N)]
def chain(s, n=20, us=20):
    for _ in range(n):
        lib.mv2d_spin(us, s.cuda_stream)
def timed(group):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in group: chain(s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for s in streams: chain(s, 2)
print('single', ' '.join(f'{timed([s]):.2f}' for s in streams))
print('pair matrix (ms; ~0.45 = concurrent, ~0.9 = serialised)')
for i in range(N):
    print(' '.join('  -- ' if i == j else f'{timed([streams[i], streams[j]]):5.2f}' for j in range(N)))
print('first 4 together', f'{timed(streams[:4]):.2f}', ' all', f'{timed(streams):.2f}')
