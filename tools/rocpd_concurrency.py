#!/usr/bin/env python
"""Concurrency of a rocprofv3 --kernel-trace database inside the window where the most queues are active (the multi-stream timed loop of bench.py): share of the
wall time with 0 / 1 / 2 / ... kernels in flight, the same for WIDE kernels (>= 256 workgroups), union busy time, and the widest kernels' share.
    python tools/rocpd_concurrency.py s_results.db [window_ms]"""
import sqlite3
import sys
from collections import defaultdict

path = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
c = sqlite3.connect(path)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute(f'pragma table_info({kd})')]
scol = [r[1] for r in c.execute(f'pragma table_info({ks})')]
name_col = 'kernel_name' if 'kernel_name' in scol else ('display_name' if 'display_name' in scol else 'name')
qcol = 'queue_id' if 'queue_id' in cols else 'stream_id'
rows = list(c.execute(f'select s.{name_col}, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.{qcol} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start'))
t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
# the window: win_ms long, placed where the number of distinct queues is largest (ties: the latest)
step = win_ms * 1e6 / 4
best = (0, t_lo)
t = t_lo
while t + win_ms * 1e6 <= t_hi:
    qs = {r[5] for r in rows if t <= r[1] < t + win_ms * 1e6}
    n = len([1 for r in rows if t <= r[1] < t + win_ms * 1e6])
    if (len(qs), n) >= best[0:1] + (0,) and len(qs) >= best[0]:
        best = (len(qs), t)
    t += step
w0, w1 = best[1], best[1] + win_ms * 1e6
sel = [r for r in rows if r[1] >= w0 and r[2] <= w1]
pts = []
for name, st, en, g, wg, q in sel:
    wide = (g // max(wg, 1)) >= 256
    pts.append((st, 1, wide)); pts.append((en, -1, wide))
pts.sort()
cur = wide = 0
last = sel[0][1]
hist, whist = defaultdict(float), defaultdict(float)
for tt, dl, w in pts:
    hist[min(cur, 6)] += tt - last; whist[min(wide, 4)] += tt - last; last = tt
    cur += dl
    if w:
        wide += dl
tot = sum(hist.values())
print(f'# window {tot / 1e6:.1f} ms with {best[0]} queues, {len(sel)} dispatches, sum of durations {sum(r[2] - r[1] for r in sel) / 1e6:.1f} ms')
print('kernels in flight -> share of the window:', {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print('WIDE kernels (>= 256 workgroups) in flight -> share:', {k: round(v / tot, 3) for k, v in sorted(whist.items())})
per = defaultdict(float)
for name, st, en, g, wg, q in sel:
    per[name.split('(')[0][:60]] += en - st
for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:8]:
    print(f'  {v / tot:6.3f} x window  {k}')
