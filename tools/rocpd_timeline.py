#!/usr/bin/env python
"""Timeline of the LAST window of a rocprofv3 --kernel-trace database: per queue, every dispatch with its start (relative), duration and
the gap to the previous dispatch on the same queue; plus the wall time covered and the union busy time.

    python tools/rocpd_timeline.py x_results.db --last-ms 8 [--from-kernel NAME] > timeline.txt
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    last_ms = float(sys.argv[sys.argv.index('--last-ms') + 1]) if '--last-ms' in sys.argv else 8.0
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute(f'pragma table_info({kd})')]
    scol = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    name_col = 'kernel_name' if 'kernel_name' in scol else ('display_name' if 'display_name' in scol else 'name')
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    q = f'select s.{name_col}, d.start, d.end, d.grid_size_x, d.workgroup_size_x, {"d." + qcol if qcol else "0"} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start'
    rows = list(c.execute(q))
    t_end = max(r[2] for r in rows)
    rows = [r for r in rows if r[1] >= t_end - last_ms * 1e6]
    t0 = rows[0][1]
    last_end = {}
    busy, cur_s, cur_e = 0, None, None
    for name, st, en, g, w, qid in rows:
        short = name.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:44]
        gap = (st - last_end[qid]) / 1e3 if qid in last_end else 0.0
        last_end[qid] = en
        print(f'q{qid:<3d} t={(st - t0) / 1e3:9.1f} us  dur={(en - st) / 1e3:7.1f}  gap={gap:7.1f}  blocks={g // max(w, 1):6d}  {short}')
        if cur_e is None or st > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = st, en
        else:
            cur_e = max(cur_e, en)
    busy += cur_e - cur_s
    print(f'# window {(rows[-1][2] - t0) / 1e6:.3f} ms, {len(rows)} dispatches, union busy {busy / 1e6:.3f} ms, sum of durations {sum(r[2] - r[1] for r in rows) / 1e6:.3f} ms')


if __name__ == '__main__':
    main()
