#!/usr/bin/env python
"""One sample per launch on 4 streams (the reference's call shape): where does the time per frame go -- host-side submission or waiting for the GPU?
    python tools/host_vs_gpu_batch1.py [workload] [streams]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mv2d_amd import ops, synthetic  # noqa: E402
from mv2d_amd.engine import HeadEngine  # noqa: E402
from mv2d_amd.streams import concurrent_streams  # noqa: E402

WL = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_s'
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
prob = synthetic.make_problem(WL, seed=0)
base = HeadEngine(synthetic.make_head_state(seed=0), prob['kind'], dev, num_views=prob['views_per_frame'])
base.fork_qg = False
engs = [base] + [base.clone_shared() for _ in range(NS - 1)]
pool = concurrent_streams(min(NS, 4), dev)
streams = [pool[i % len(pool)] for i in range(NS)]
feats = [torch.randn(prob['feat'].shape, device=dev) for _ in range(NS)]
props = [torch.from_numpy(p) for p in prob['proposals']]
pay = torch.zeros((NS, 3301), device=dev)
waited = [0.0]
orig = torch.cuda.Event.synchronize


def timed_sync(self):
    t = time.perf_counter()
    orig(self)
    waited[0] += time.perf_counter() - t


torch.cuda.Event.synchronize = timed_sync


def step():
    for i, (e, s) in enumerate(zip(engs, streams)):
        with torch.cuda.stream(s):
            o = e.run(feats[i], props, prob['img_metas'], use_graph=True)
            ops.pack_detections(o['boxes'], o['scores'], o['labels'], o['count'], pay[i:i + 1])


for _ in range(10):
    step()
torch.cuda.synchronize()
waited[0] = 0.0
N = 300
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
fr = N * NS
print(f'{WL}, {NS} streams x 1 sample: {fr / (t2 - t0):.0f} samples/s; per frame: wall {1e6 * (t2 - t0) / fr:.0f} us, host loop {1e6 * (t1 - t0) / fr:.0f} us of which '
      f'{1e6 * waited[0] / fr:.0f} us waiting for the staging event -> host work {1e6 * (t1 - t0 - waited[0]) / fr:.0f} us')
