import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import synthetic, dist as mdist
from mv2d_amd.engine import HeadEngine
dev = torch.device('cuda:0')
prob = synthetic.make_problem('cfg2_s', seed=0)
sd = synthetic.make_head_state(seed=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
base = HeadEngine(sd, 'S', dev, num_views=6)
engines = [base] + [base.clone_shared() for _ in range(n - 1)]
streams = [torch.cuda.Stream() for _ in range(n)]
feat = torch.from_numpy(prob['feat']).to(dev)
props = [torch.from_numpy(p) for p in prob['proposals']]
metas = prob['img_metas']
def step(pack=True):
    cur = torch.cuda.current_stream()
    for e, s in zip(engines, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            o = e.run(feat, props, metas, use_graph=True)
            if pack:
                mdist.pack_detections(o['boxes'], o['scores'], o['labels'], o['count'])
        cur.wait_stream(s)
for _ in range(5): step()
torch.cuda.synchronize()
for pack in (True, False):
    t0 = time.perf_counter()
    for _ in range(30): step(pack)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'pack={pack}: enqueue {1e3*(t1-t0)/30/n:.3f} ms/frame, total {1e3*(t2-t0)/30/n:.3f} ms/frame')
pr = cProfile.Profile(); pr.enable()
for _ in range(30): step(False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
