#!/usr/bin/env python
"""Replay the head engine's hipGraph a few times on one stream (a profiling target for tools/prof_cmd.sh):
    python tools/run_engine.py --workload cfg2_s --batch 8 --steps 20 [--exact] [--eager]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mv2d_amd import synthetic  # noqa: E402
from mv2d_amd.engine import HeadEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='cfg2_s')
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--exact', action='store_true', help='(the default since round 5; accepted and ignored)')
ap.add_argument('--key16', action='store_true', help='the opt-in key16 mode (one fp16 rounding of the key side)')
ap.add_argument('--eager', action='store_true')
ap.add_argument('--pe-v2', action='store_true', help='the opt-in second shape of the PE kernel (csrc/pe_x3b.hip) instead of csrc/pe_x3.hip')
ap.add_argument('--group', type=int, default=None, help='1 / 0: force the shared-tile cross attention (csrc/xattn_group.hip) on / off; default: the engine chooses (T path: on)')
a = ap.parse_args()
dev = torch.device('cuda:0')
probs = [synthetic.make_problem(a.workload, seed=s) for s in range(a.batch)]
eng = HeadEngine(synthetic.make_head_state(seed=0), probs[0]['kind'], dev, num_views=probs[0]['views_per_frame'], exact=not a.key16)
eng.fork_qg = False
if a.pe_v2:
    eng.pe_rows_in_waves = True
if a.group is not None:
    eng.group_xattn = bool(a.group)
feats = torch.cat([torch.from_numpy(p['feat']) for p in probs]).to(dev)
props = [[torch.from_numpy(x) for x in p['proposals']] for p in probs]
metas = [p['img_metas'] for p in probs]
run = (lambda: eng.run_batch(feats, props, metas, use_graph=not a.eager)) if a.batch > 1 else (lambda: eng.run(feats, props[0], metas[0], use_graph=not a.eager))
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f'{a.workload} batch {a.batch} route={"key16" if a.key16 else "index-exact"}: {dt * 1e3:.3f} ms per launch sequence, {a.batch / dt:.0f} samples/s on one stream')
