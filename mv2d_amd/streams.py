"""Stream planner: pick HIP streams whose hardware queues really execute side by side.

On MI355X a process owns a handful of HSA hardware queues (GPU_MAX_HW_QUEUES, default 4) that the compute micro-engine serves
from 4 pipes.  HIP streams are multiplexed onto them in creation order, so two streams picked at random frequently share a
queue or a pipe and their kernels then run strictly one after the other (measured: the same 4 frames in flight give 1000, 1500
or 2100 samples/s depending on nothing but which streams they happened to get).  ``concurrent_streams`` measures instead of
guessing: candidates are probed pairwise with short chains of a one-wave spin kernel (mv2d_spin) and a set of mutually concurrent
streams is returned.  Costs a few tens of milliseconds once per process.
"""
import time

import torch

from . import _lib

_cache = {}


def _chain(lib, stream, n, usec):
    for _ in range(n):
        lib.mv2d_spin(usec, stream.cuda_stream)


def _timed(lib, group, n, usec, device):
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for s in group:
        _chain(lib, s, n, usec)
    torch.cuda.synchronize(device)
    return time.perf_counter() - t0


def concurrent_streams(n=4, device=None, candidates=16, chain=10, usec=40):
    """Up to ``n`` torch streams that pairwise overlap (greedy clique over ``candidates`` fresh streams).  Always returns at
    least one stream; fewer than ``n`` when the device exposes fewer independent queues."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    key = (device.index, n)
    if key in _cache:
        return _cache[key]
    lib = _lib.load()
    cands = [torch.cuda.Stream(device=device) for _ in range(candidates)]
    for s in cands:
        _chain(lib, s, 2, 5)                       # bind every stream to its hardware queue before measuring
    single = min(_timed(lib, [cands[0]], chain, usec, device) for _ in range(2))
    chosen = [cands[0]]
    for c in cands[1:]:
        if len(chosen) >= n:
            break
        if all(min(_timed(lib, [x, c], chain, usec, device) for _ in range(2)) < 1.5 * single for x in chosen):
            chosen.append(c)
    _cache[key] = chosen
    return chosen
