"""A SECOND, independently written statement of mmcv 1.6.1 ``RoIAlign`` (aligned=True, pool_mode='avg', sampling_ratio=-1,
output 7x7) — test infrastructure only.

mmcv is a third-party dependency that is absent from /root/reference, so ``oracle.mv2d_oracle.roi_align`` (which the stub layer
of oracle/gen_golden.py also uses) is pinned by no reference-side test.  To take the circularity out of "the oracle agrees with
a golden that the oracle produced", this file states the same published semantics a different way — dense sampling through
``torch.nn.functional.grid_sample`` instead of explicit four-tap gathers — written from the specification (SURVEY.md appendix A2,
mmcv/ops/csrc/pytorch/cpu/roi_align.cpp's documented behaviour) WITHOUT consulting oracle/mv2d_oracle.py.
tests/test_roi_align_cross_cpu.py checks the two against each other.

Specification used:
  * roi = (batch index, x1, y1, x2, y2) in image pixels; feature coordinates = pixel * spatial_scale - 0.5 (aligned=True);
  * bin size = roi extent / 7 (no clamping of the extent when aligned); samples per bin per axis = ceil(roi extent / 7);
  * sample s of bin p at  start + p * bin + (s + 0.5) * bin / samples;
  * a sample with y < -1 or y > H or x < -1 or x > W contributes 0; otherwise coordinates are clamped to [0, size - 1] and the
    value is the bilinear interpolation between the integer neighbours (pixel centres at integers);
  * the bin value is the sum over its samples / max(samples_y * samples_x, 1).
grid_sample with align_corners=True places pixel centres at integer coordinates and padding_mode='border' clamps the sampling
position to [0, size - 1] — exactly the clamp above — so only the validity window has to be applied by hand.
"""
import math

import torch
import torch.nn.functional as F


def roi_align_grid_sample(feat, rois, out_size=7, spatial_scale=1.0 / 16):
    """feat [N,C,H,W] float, rois [R,5] -> [R,C,out,out] (computed in float64)."""
    feat = feat.double()
    N, C, H, W = feat.shape
    out = feat.new_zeros((rois.shape[0], C, out_size, out_size))
    for i, roi in enumerate(rois.double()):
        n = int(roi[0].item())
        x_lo, y_lo, x_hi, y_hi = [(v * spatial_scale - 0.5).item() for v in roi[1:]]
        rw, rh = x_hi - x_lo, y_hi - y_lo
        bw, bh = rw / out_size, rh / out_size
        sx, sy = int(math.ceil(rw / out_size)), int(math.ceil(rh / out_size))
        if sx <= 0 or sy <= 0:
            continue                                       # no samples: the bins stay 0 (sum of nothing / max(0, 1))
        # all sample positions of the RoI: a (out*sy) x (out*sx) lattice
        ys = y_lo + (torch.arange(out_size * sy, dtype=torch.float64) + 0.5) * (bh / sy)
        xs = x_lo + (torch.arange(out_size * sx, dtype=torch.float64) + 0.5) * (bw / sx)
        ok_y = (ys >= -1.0) & (ys <= H)
        ok_x = (xs >= -1.0) & (xs <= W)
        gy = (2.0 * ys / (H - 1) - 1.0) if H > 1 else torch.zeros_like(ys)
        gx = (2.0 * xs / (W - 1) - 1.0) if W > 1 else torch.zeros_like(xs)
        grid = torch.stack(torch.meshgrid(gy, gx, indexing='ij')[::-1], -1)[None]            # [1, Y, X, (x, y)]
        vals = F.grid_sample(feat[n:n + 1], grid, mode='bilinear', padding_mode='border', align_corners=True)[0]   # [C, Y, X]
        vals = vals * (ok_y[:, None] & ok_x[None, :])
        out[i] = vals.reshape(C, out_size, sy, out_size, sx).sum((2, 4)) / max(sy * sx, 1)
    return out
