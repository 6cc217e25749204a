"""The extra FPN level (1x1 lateral + 3x3 conv) on the cfg-2 map: time in a hipGraph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mv2d_amd
dev = torch.device('cuda:0')
neck = mv2d_amd.build_neck(dict(type='FPN', in_channels=[256] * 5, out_channels=256, start_level=2, end_level=2, num_outs=1)).to(dev)
for V, h, w in ((6, 32, 88), (12, 40, 100)):
    feats = [None, None, torch.randn(V, 256, h, w, device=dev), None, None]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): neck(feats)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): out = neck(feats)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    P = V * h * w
    fl = 2.0 * P * 256 * (256 + 2304)
    print(f'V={V} {h}x{w}: {us:.1f} us per neck pass (transpose+cast, 1x1, 3x3), {fl / us / 1e6:.0f} TFLOP/s')
