import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import synthetic
from mv2d_amd.engine import HeadEngine
dev = torch.device('cuda:0')
prob = synthetic.make_problem('cfg2_s', seed=0)
sd = synthetic.make_head_state(seed=0)
base = HeadEngine(sd, 'S', dev, num_views=6)
feat = torch.from_numpy(prob['feat']).to(dev)
props = [torch.from_numpy(p) for p in prob['proposals']]
metas = prob['img_metas']
NS = 8
engines = [base] + [base.clone_shared() for _ in range(NS - 1)]
outs = [e.run(feat, props, metas) for e in engines]
torch.cuda.synchronize()
R = outs[0]['R']
def capture(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return g
def bench(name, graphs):
    for ns in (1, 2, 4, 8):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        def run(reps):
            for _ in range(reps):
                for g, s in zip(graphs[:ns], streams):
                    with torch.cuda.stream(s): g.replay()
        run(3); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(30); torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f'{name}: streams={ns}: {1e3*(t1-t0)/(30*ns):.3f} ms per graph aggregate')
dec = [capture(lambda e=e, o=o: e._enqueue_decoder(o['ws'], R)) for e, o in zip(engines, outs)]
bench('decoder', dec)
from mv2d_amd import ops
def pe(e, o):
    ws, W_ = o['ws'], e.w
    md = ws['S_dev']
    ops.gemm_bf16(ws['A1'], W_['pe_w1a'], W_['pe_b1a'], m_dev=md, act=1, out=ws['H1'])
    ops.gemm_bf16(ws['A2'], W_['pe_w2a'], W_['pe_b2a'], m_dev=md, act=1, out=ws['H2'])
    ops.gemm_bf16(ws['H1'], W_['pe_w1b'], W_['pe_b1b'], m_dev=md, mul=ws['gate'], out=ws['Pg'])
    ops.gemm_bf16(ws['roi_sum'].view(R * 49, 256), W_['kv_w'], W_['kv_b'], A2=ws['roi_feat'].view(R * 49, 256), n_split=6 * 256,
                  out=ws['KV'], ldc=256, c_blk_stride=ws['S_kv'] * 256, c_blk_cols=256)
big = [capture(lambda e=e, o=o: pe(e, o)) for e, o in zip(engines, outs)]
bench('big_gemms', big)

# mixed: decoder graph on one stream + big GEMM graph on another (different engines' buffers)
def mixed(nd, nb):
    sd_ = [torch.cuda.Stream() for _ in range(nd)]; sb_ = [torch.cuda.Stream() for _ in range(nb)]
    def run(reps):
        for _ in range(reps):
            for g, s in zip(dec[:nd], sd_):
                with torch.cuda.stream(s): g.replay()
            for g, s in zip(big[4:4 + nb], sb_):
                with torch.cuda.stream(s):
                    for _ in range(4): g.replay()
    run(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(30); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'mixed: {nd} decoder graph(s) + {nb} x4 big-gemm graph(s) per round: {1e3*(t1-t0)/30:.3f} ms per round')
mixed(1, 0); mixed(0, 1); mixed(1, 1); mixed(2, 2)
