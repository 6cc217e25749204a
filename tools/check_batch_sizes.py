"""run_batch at larger batch sizes (8 x cfg2_s, 3 x cfg5_t, 6 x cfg3_t, hipGraph) == single-sample runs, bitwise; peak memory."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import synthetic
from mv2d_amd.engine import HeadEngine
dev = torch.device('cuda:0')
sd = synthetic.make_head_state(seed=0)
for name, B in (('cfg2_s', 8), ('cfg5_t', 3), ('cfg3_t', 6)):
    probs = [synthetic.make_problem(name, seed=7 * b + 1) for b in range(B)]
    kind, vpf = probs[0]['kind'], probs[0]['views_per_frame']
    eng = HeadEngine(sd, kind, dev, num_views=vpf)
    feats = [torch.from_numpy(p['feat']).to(dev) for p in probs]
    props = [[torch.from_numpy(x) for x in p['proposals']] for p in probs]
    metas = [p['img_metas'] for p in probs]
    out = eng.run_batch(torch.cat(feats, 0), props, metas, use_graph=True)
    res = eng.results_batch(out)
    grp = out['grp_start'].tolist()
    single = HeadEngine(sd, kind, dev, num_views=vpf)
    ok = True
    for b in range(B):
        ref = single.run(feats[b], props[b], metas[b])
        ok &= torch.equal(out['cls'][:, grp[b]:grp[b + 1]], ref['cls']) and all(torch.equal(x, y) for x, y in zip(res[b], single.results(ref)))
    print(name, 'B', B, 'R', out['R'], 'bitwise equal to single runs:', ok, 'mem GB', round(torch.cuda.max_memory_allocated() / 2**30, 2))
