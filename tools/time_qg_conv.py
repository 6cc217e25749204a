"""Time mv2d_qg_conv_pool for both block shapes: python tools/time_qg_conv.py [R]"""
import os, sys, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    from mv2d_amd import ops
    R = int(sys.argv[1]); dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(R, 49, 256, generator=g)).to(dev).to(torch.bfloat16)
    w = (torch.randn(256, 2304, generator=g) * 0.03).to(dev).to(torch.bfloat16)
    b = torch.randn(256, generator=g).to(dev)
    wp = ops.pack_wfrag(w); out = torch.empty((R, 256), device=dev)
    for _ in range(5): ops.qg_conv_pool(x, wp, b, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.qg_conv_pool(x, wp, b, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 2.0 * R * 49 * 2304 * 256
    print('NR=%s R=%d: %.1f us  %.0f TFLOP/s (%.3f of 2500)  checksum %.6f' % (os.environ.get('MV2D_QG_CONV_NR'), R, us, fl / us / 1e6, fl / us / 1e6 / 2500, float(out.double().sum())))
else:
    R = sys.argv[1] if len(sys.argv) > 1 else '2400'
    for nr in ('1', '2'):
        subprocess.run([sys.executable, __file__, R, 'child'], env=dict(os.environ, MV2D_QG_CONV_NR=nr))
