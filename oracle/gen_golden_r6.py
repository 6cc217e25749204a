"""Round-6 goldens from the UNMODIFIED reference (build container only):

    python -B -m oracle.gen_golden_r6          # writes tests/golden/{cfg2_s_nc6,cfg3_t}_seed2.npz, cfg2_s_r450.npz, cfg2_s_dup.npz, cfg5_t_dup.npz,
                                               # and the reference-against-itself rank noise of each into tests/golden/refnoise.npz

* a THIRD seed of the overlapping S rig and of the two-frame T workload (seed 2);
* the S path at the cap of SURVEY 8.0 (75 boxes per view, R = 450);
* near-duplicate 2-D boxes (mv2d_amd.synthetic: a fifth of every view's boxes are copies, half exact, half 0.25 px larger) at the headline S size on the
  overlapping rig and at the cfg-5 T size -- ties and near-ties in the IoU ranking of the box correlation (RH/utils/box_correlation.py:370-374).
Every case is run under the execution variants of oracle/gen_golden_refnoise.py (intra-op threads 8 / 1 / 16 / 4, oneDNN off): the t8 run is the golden,
the pairwise differing-rank counts go into refnoise.npz under <name>_s<seed> (the bound tests/test_gpu_golden.py holds the HIP path to).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from mv2d_amd import synthetic  # noqa: E402
from oracle import _stubs  # noqa: E402
from oracle.gen_golden import build_reference_head, run_case, OUT  # noqa: E402
from oracle.gen_golden_refnoise import VARIANTS, ranked_diff  # noqa: E402

CASES = [('cfg2_s_nc6', 2), ('cfg3_t', 2), ('cfg2_s_r450', 0), ('cfg2_s_dup', 0), ('cfg5_t_dup', 0)]


def main():
    S_cls, T_cls = _stubs.install('/root/reference')
    only = [a for a in sys.argv[1:] if not a.startswith('-')]
    sd_np = synthetic.make_head_state(seed=0)
    path = os.path.join(OUT, 'refnoise.npz')
    store = dict(np.load(path)) if os.path.exists(path) else {}
    for name, seed in CASES:
        if only and name not in only:
            continue
        prob = synthetic.make_problem(name, seed=seed)
        recs = {}
        for vname, v in VARIANTS:
            torch.set_num_threads(v['threads'])
            torch.backends.mkldnn.enabled = v['mkldnn']
            head = build_reference_head(prob['kind'], S_cls, T_cls, sd_np, prob['views_per_frame'])
            recs[vname] = run_case(head, prob['kind'], prob['feat'], prob['proposals'], prob['img_metas'], False)
        torch.backends.mkldnn.enabled = True
        base = recs['t8']
        key = f'{name}_s{seed}'
        n = len(VARIANTS)
        pair = np.zeros((n, n), np.int32)
        for i, (vi, _) in enumerate(VARIANTS):
            for j, (vj, _) in enumerate(VARIANTS):
                pair[i, j] = ranked_diff(recs[vi]['topk_index'], recs[vj]['topk_index'])
        store[key + '_variants'] = np.array([v for v, _ in VARIANTS])
        store[key + '_topk_index'] = np.stack([recs[v]['topk_index'] for v, _ in VARIANTS])
        store[key + '_topk_scores'] = np.stack([recs[v]['topk_scores'] for v, _ in VARIANTS])
        store[key + '_pairwise_ranked_diff'] = pair
        gaps = [0.0]
        pos = {int(x): j for j, x in enumerate(base['topk_index'])}
        for v, _ in VARIANTS[1:]:
            for i, x in enumerate(recs[v]['topk_index']):
                j = pos.get(int(x))
                if j is not None and j != i:
                    gaps.append(abs(float(base['topk_scores'][i]) - float(base['topk_scores'][j])))
        store[key + '_max_tie_gap'] = np.float64(max(gaps))
        store[key + '_cls_dev'] = np.float64(max(float(np.abs(recs[v]['cls'] - base['cls']).max()) for v, _ in VARIANTS[1:]) / float(np.abs(base['cls']).max()))
        out = name + ('.npz' if seed == 0 else f'_seed{seed}.npz')
        np.savez_compressed(os.path.join(OUT, out), **base)
        np.savez_compressed(path, **store)
        print(key, '->', out, 'R', base['cls'].shape[1], 'rank noise of the reference against itself: max', int(pair.max()), 'largest gap crossed %.2e' % max(gaps),
              'corr' if 'corr' in base else '', base['corr'].shape if 'corr' in base else '', flush=True)


if __name__ == '__main__':
    main()
