"""host-side cost of submitting one launch (run_batch with hipGraph replay), vs the GPU time of the launch: python tools/host_time.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv2d_amd import synthetic
from mv2d_amd.engine import HeadEngine
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
WL = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_s'
dev = torch.device('cuda:0')
probs = [synthetic.make_problem(WL, seed=s) for s in range(B)]
eng = HeadEngine(synthetic.make_head_state(seed=0), probs[0]['kind'], dev, num_views=probs[0]['views_per_frame'])
feats = torch.cat([torch.as_tensor(p['feat']).to(dev) for p in probs])
props = [[torch.as_tensor(q) for q in p['proposals']] for p in probs]
metas = [p['img_metas'] for p in probs]
for _ in range(5):
    out = eng.run_batch(feats, props, metas, use_graph=True)
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    out = eng.run_batch(feats, props, metas, use_graph=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host submit %.3f ms per launch; with GPU drain %.3f ms per launch' % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    out = eng.run_batch(feats, props, metas, use_graph=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(int(os.environ.get('ROWS', '18')))
