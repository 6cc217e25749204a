#!/usr/bin/env python
"""Per-kernel average of the counters collected by `rocprofv3 --kernel-trace --pmc <COUNTER>` (rocpd SQLite output).

    python tools/rocpd_pmc.py gpurun_out/pmc_FETCH_SIZE/p_results.db [--by-grid]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  NOTE (MI355X_MICROARCH.md §HBM): on gfx950 FETCH_SIZE counts wide
coalesced reads at half their size -> double it before comparing with a byte count; WRITE_SIZE is uncalibrated.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = '--by-grid' in sys.argv
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pe, ip, kd, ks = t('rocpd_pmc_event'), t('rocpd_info_pmc'), t('rocpd_kernel_dispatch'), t('rocpd_info_kernel_symbol')
    scol = [r[1] for r in db.execute(f'pragma table_info({ks})')]
    name_col = 'kernel_name' if 'kernel_name' in scol else 'display_name'
    q = (f'select s.{name_col}, d.grid_size_x, d.workgroup_size_x, d.grid_size_y, d.grid_size_z, i.name, e.value '
         f'from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id join {ip} i on e.pmc_id = i.id')
    agg = {}
    for name, gx, wx, gy, gz, cname, val in db.execute(q):
        short = name.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')
        key = (short, (gx // max(wx, 1), gy, gz) if by_grid else (), cname)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(val)
    print(f'{"kernel":60s} {"blocks":>14s} {"counter":>12s} {"calls":>6s} {"avg":>14s} {"total":>16s}')
    for (k, g, c), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{k[:60]:60s} {"x".join(map(str, g)):>14s} {c:>12s} {n:6d} {tot / n:14.1f} {tot:16.1f}')


if __name__ == '__main__':
    main()
