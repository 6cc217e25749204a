#!/usr/bin/env python
"""Which key-side rounding of the default route costs how many ranks?  The index-exact route with single stages downgraded to the default
route's one key16 rounding (HeadEngine.exact_skip), against the reference goldens:
    python tools/ablate_exact.py [workloads...]      (default: cfg2_s cfg3_t cfg5_t cfg2_s_nc6)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mv2d_amd import synthetic  # noqa: E402
from mv2d_amd.engine import HeadEngine  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device('cuda:0')
sd = synthetic.make_head_state(seed=0)
for name in (sys.argv[1:] or ['cfg2_s', 'cfg3_t', 'cfg5_t', 'cfg2_s_nc6']):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    prob = synthetic.make_problem(name, seed=0)
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    variants = [('default route', None), ('exact, all stages', frozenset()), ('exact minus attn', frozenset({'attn'})), ('exact minus pe', frozenset({'pe'})),
                ('exact minus conv', frozenset({'conv'})), ('exact: attn only', frozenset({'pe', 'conv'})), ('exact: pe only', frozenset({'attn', 'conv'})),
                ('exact: conv only', frozenset({'attn', 'pe'})), ('exact, value rows hi only', 'zero_v'), ('exact, key rows hi only', 'zero_k'), ('exact, lo rows as e4m3', 'f8z'), ('exact, lo rows as e5m2', 'f8')]
    for label, skip in variants:
        eng = HeadEngine(sd, prob['kind'], dev, num_views=prob['views_per_frame'], exact=skip is not None)
        if isinstance(skip, str):
            eng.lo8_rows = False                                     # (the emulations rewrite key16 lo rows)
            eng.ablate_zero_lo = frozenset({'8', '8z'}) if skip == 'f8z' else frozenset({'8'}) if skip == 'f8' else frozenset({skip[-1]})
        elif skip is not None:
            eng.exact_skip = skip
        out = eng.run(feat, props, prob['img_metas'])
        torch.cuda.synchronize()
        n = int(out['count'].item())
        R = out['R']
        flat = out['bbox_index'][:n].cpu().numpy() * 10 + out['labels'][:n].cpu().numpy()
        ref = g['topk_index']
        m = min(n, len(ref))
        cls = out['cls'][:, :R].cpu().numpy().reshape(g['cls'].shape)
        e_cls = float(np.abs(cls - g['cls']).max() / np.abs(g['cls']).max())
        e_sc = float(np.abs(out['scores'][:m].cpu().numpy() - g['scores'][:m]).max())
        print(f'{name:11s} {label:20s}: ranked indices differing {int((flat[:m] != ref[:m]).sum()) + abs(n - len(ref)):3d}/{len(ref)}, cls rel err {e_cls:.1e}, score err {e_sc:.1e}')
