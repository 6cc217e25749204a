"""CPU baseline leg of bench.py: time the oracle (a port of the reference algorithm, torch CPU fp32/fp64) on a bounded
sample of the bench workload.  Run as a subprocess so that a slow host cannot hang the benchmark.

    python -m oracle.cpu_baseline --workload cfg2_s --iters 3 --threads 16   ->  one JSON line
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='cfg2_s')
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--warmups', type=int, default=1, help='untimed frames first (BASELINE.md section 3 protocol: 3)')
    ap.add_argument('--budget-s', type=float, default=0.0, help='> 0: stop warming up after half of it and stop timing after all of it (at least one '
                                                               'timed frame is always taken): a bounded leg that reports what it measured')
    ap.add_argument('--no-decoder', action='store_true', help='skip the decoder-only leg')
    a = ap.parse_args()
    import torch
    from mv2d_amd import synthetic
    from oracle import mv2d_oracle as O
    torch.set_num_threads(a.threads)
    prob = synthetic.make_problem(a.workload, seed=0)
    sd = synthetic.make_head_state(seed=0)
    fn = O.forward_t if prob['kind'] == 'T' else O.forward_s
    kw = {'num_views': prob['views_per_frame']} if prob['kind'] == 'T' else {}
    feat = torch.from_numpy(prob['feat'])
    props = [torch.from_numpy(p) for p in prob['proposals']]
    ts = []
    t_start = time.perf_counter()
    over = lambda frac: a.budget_s > 0 and time.perf_counter() - t_start > frac * a.budget_s      # noqa: E731
    warm = 0
    with torch.no_grad():
        prog = lambda phase, t: print(json.dumps(dict(progress=phase, frame_s=round(t, 3), threads=a.threads)), flush=True)      # noqa: E731
        for _ in range(max(a.warmups, 1)):
            t0 = time.perf_counter()
            fn(sd, feat, props, prob['img_metas'], **kw)
            warm += 1
            prog('warmup', time.perf_counter() - t0)          # (one line per frame: a caller that gives up on this leg still sees what it measured)
            if over(0.5):
                break
        for _ in range(a.iters):
            t0 = time.perf_counter()
            fn(sd, feat, props, prob['img_metas'], **kw)
            ts.append(time.perf_counter() - t0)
            prog('timed', ts[-1])
            if over(1.0):
                break
        td = [float('nan')]
        if a.no_decoder:
            return report(a, torch, ts, td, warm)
        # decoder-only leg (CrossAttentionBoxHead.forward on prepared inputs: 6 layers + heads), the second headline metric
        st = {}
        fn(sd, feat, props, prob['img_metas'], stages=st, **kw)
        sdt = O._to_t(sd)
        if prob['kind'] == 'T':
            mem = feat.permute(0, 2, 3, 1)[st['roi_mask']]
            mpe = st['pe'].permute(0, 2, 3, 1)[st['roi_mask']]
            dec = lambda: O.pred_heads(sdt, O.decoder(sdt, st['qpos'], mem + mpe, mem, st['blocked']), st['ref'])
        else:
            rf = st['roi_feats'].flatten(2).transpose(1, 2)
            rp = st['roi_pe'].flatten(2).transpose(1, 2)
            dec = lambda: O.pred_heads(sdt, O.decoder_s(sdt, st['qpos'], rf + rp, rf, st['corr'], st['corr_mask']), st['ref'])
        dec()
        td = []
        for _ in range(max(3, a.iters // 4)):
            t0 = time.perf_counter()
            dec()
            td.append(time.perf_counter() - t0)
    report(a, torch, ts, td, warm)


def report(a, torch, ts, td, warm):
    med = statistics.median(ts)
    dec = statistics.median(td)
    print(json.dumps(dict(value=round(1.0 / med, 4), unit='samples/s', cores=a.threads, kind='port',
                          decoder_ms_per_iter=None if dec != dec else round(dec * 1e3, 2), frames_timed=len(ts), warmups=warm,
                          sample=f'{len(ts)} frames of {a.workload} after {warm} warm-up(s), median {med * 1e3:.0f} ms/frame, torch '
                                 f'{torch.__version__} CPU fp32/fp64, {a.threads} threads of {os.cpu_count()} host cores' +
                                 (f' (bounded to {a.budget_s:.0f} s)' if a.budget_s > 0 else ''))))


if __name__ == '__main__':
    main()
