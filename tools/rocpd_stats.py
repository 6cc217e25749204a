#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace rocpd SQLite database into a per-kernel table (count / total / avg / min /
max duration), optionally split by launch grid so that the different shapes of one GEMM kernel are separated.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--by-grid] [--skip N] > profiles/r01_xxx.txt
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    by_grid = '--by-grid' in sys.argv
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute(f'pragma table_info({kd})')]
    scol = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    name_col = 'kernel_name' if 'kernel_name' in scol else ('display_name' if 'display_name' in scol else 'name')
    gx = 'grid_size_x' if 'grid_size_x' in cols else 'grid_x'
    wx = 'workgroup_size_x' if 'workgroup_size_x' in cols else 'workgroup_x'
    q = f'select s.{name_col}, d.start, d.end, d.{gx}, d.grid_size_y, d.grid_size_z, d.{wx} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start'
    rows = list(c.execute(q))
    agg = {}
    for name, st, en, g0, g1, g2, w0 in rows:
        short = name.split('(')[0]
        for pre in ('void ', '(anonymous namespace)::'):
            short = short.replace(pre, '')
        key = (short, (g0 // max(w0, 1), g1, g2)) if by_grid else (short,)
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print(f'# {path}: {len(rows)} dispatches, {tot / 1e6:.3f} ms total kernel time')
    print(f'{"kernel":64s} {"blocks":>16s} {"calls":>7s} {"total_us":>11s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} {"pct":>6s}')
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        grid = 'x'.join(str(v) for v in key[1]) if by_grid else ''
        print(f'{key[0][:64]:64s} {grid:>16s} {a[0]:7d} {a[1] / 1e3:11.1f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100.0 * a[1] / tot:6.2f}')


if __name__ == '__main__':
    main()
