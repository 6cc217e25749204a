"""per-phase s_memtime stamps of one block of the one-launch cross attention (csrc/xattn_fused.hip) on the cfg2_s frame of the engine; variant library:
   tools/build_variant.sh xftrace xattn_fused.hip -DMV2D_XF_TRACE=300 && MV2D_HIP_LIB=mv2d_amd/lib/variants/libxftrace.so python tools/xf_trace.py [workload]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mv2d_amd import ops, synthetic  # noqa: E402
from mv2d_amd.engine import HeadEngine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_s'
B = 16
dev = torch.device('cuda:0')
probs = [synthetic.make_problem(name, seed=s) for s in range(B)]
eng = HeadEngine(synthetic.make_head_state(seed=0), probs[0]['kind'], dev, num_views=probs[0]['views_per_frame'])
feats = torch.cat([torch.from_numpy(p['feat']) for p in probs]).to(dev)
out = eng.run_batch(feats, [[torch.from_numpy(x) for x in p['proposals']] for p in probs], [p['img_metas'] for p in probs])
torch.cuda.synchronize()
ws, W_ = out['ws'], eng.w
R = ws['x'].shape[0]
fn = lambda: ops.xattn_fused(ws['q'], W_['ca_mapA0'], W_['ca_mapB0'], W_['ca_v_b0'], ws['xk_rows'], ws['xv_rows'], ws['row_ptr'], ws['col_idx'], out=ws['ctx'], R=R,      # noqa: E731
                             order=ws['q_order'], Xk_lo=ws['xk_lo'], Xv_lo=ws['xv_lo'])
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
print(f'xattn_fused {name} x {B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch')
lib = ctypes.CDLL(os.environ['MV2D_HIP_LIB'])
if hasattr(lib, 'mv2d_xf_trace_read'):
    buf = (ctypes.c_longlong * 32)()
    lib.mv2d_xf_trace_read(buf, 32)
    t = list(buf)
    names = {1: 'slots', 2: 'phase A (query maps) + barrier', 3: 'tile 0: loads -> LDS', 4: 'tile 0: compute', 5: 'tile 1: loads', 6: 'tile 1: compute', 7: 'tile 2: loads', 8: 'tile 2: compute',
             9: 'tile 3: loads', 10: 'tile 3: compute', 15: 'z -> LDS', 16: 'barrier (longest row of the block)', 17: 'phase C (context maps) + stores'}
    prev = t[0]
    for i in sorted(names):
        if t[i]:
            print(f'  {names[i]:40s} {t[i] - prev:8d}')
            prev = t[i]
    print('  total', t[17] - t[0], '(s_memtime ticks: 100 MHz)')
    print('  inside phase A (from the slot table): weight + query loads issued', t[20] - t[1], '| first two fragments used', t[21] - t[1], '| half', t[22] - t[1], '| all MFMAs issued', t[23] - t[1], '| barrier passed', t[2] - t[1])
