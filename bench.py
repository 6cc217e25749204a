#!/usr/bin/env python
"""bench.py — MV2D RoI-head hot path on MI355X: multi-view samples/s + decoder ms/iter (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (PE -> RoI gather -> query generator -> box correlation -> key list/CSR
-> K/V projection -> 6-layer decoder -> heads -> top-k decode, + the all-gather of decoded boxes when N > 1) over
one batch of `--inflight` synthetic 6-camera frames per GPU (each frame = one independent sample on its own HIP
stream, replayed as a hipGraph).  Inputs are resident in HBM before the timed region.  Prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


SINE_TABLE = os.environ.get('MV2D_PE_SINE_TABLE', '0') == '1'      # opt-in experiment: the sine branch of the PE block from a table


def stage_flops(kind, R, S, L=6):
    """Algorithmic FLOPs (2 x MAC) of the dense bf16 MFMA launches of one frame."""
    C = 256
    sine = 0 if SINE_TABLE else 384 * 1024 + 1024 * 256
    return {
        'pe_fused': 2.0 * S * (192 * 1024 + 1024 * 256 + 2 * 256 * 256 + sine),                # 2.49 MFLOP per key position
        'qg_conv_gemm': 2.0 * R * 49 * 2304 * 256,
        'kv_gemm': 2.0 * (S if kind == 'T' else R * 49) * C * (2 * L * C),
    }


def stage_bytes(kind, R, S, L=6):
    """Algorithmic HBM bytes of the same launches: A read once + weights once + output written once (bf16 = 2 B, fp32 = 4 B)."""
    C = 256
    Mkv = S if kind == 'T' else R * 49
    return {
        # inputs (A1, A2, Xf bf16, Xf fp32) + the six weight matrices once + pe fp32 and Xk bf16 out
        'pe_fused': (S * (192 + 256) * 2 + S * 256 * 4 * 2 + (192 * 1024 + 1024 * 256 + 2 * 256 * 256) * 2 + S * 256 * 6) if SINE_TABLE else
                    (S * (192 + 384 + 256) * 2 + S * 256 * 4 + (192 * 1024 + 384 * 1024 + 2 * 1024 * 256 + 2 * 256 * 256) * 2 + S * 256 * 6),
        'qg_conv_gemm': R * 49 * 256 * 2 + 2304 * 256 * 2 + R * 256 * 4,                       # pooled [R,256] output
        'kv_gemm': 2 * Mkv * C * 2 + 2 * L * C * C * 2 + Mkv * 2 * L * C * 2,
    }


STAGE_KERNEL = {'pe_fused': 'pe_fused_kernel (3 two-layer MLPs + gate + sum)', 'kv_gemm': 'kvproj_kernel', 'qg_conv_gemm': 'roi_conv_pool_kernel (conv3x3 + ReLU + avgpool fused)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='cfg2_s', help='cfg2_s (MV2D-S 6 cams 1408x512, headline) | cfg3_t | cfg5_t | cfg1_s ...')
    ap.add_argument('--inflight', type=int, default=4, help='HIP streams per GPU, each running its own launch sequence per step')
    ap.add_argument('--batch', type=int, default=8, help='samples sharing every launch of a stream (HeadEngine.run_batch)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL; gloo only for single-GPU dry runs)')
    ap.add_argument('--corr-topk', type=int, default=None)    # S path: correlated RoIs per other view (reference default 1); T path: 20
    ap.add_argument('--force-nc', type=int, default=None)     # S path sweep: synthetic correlation lists, n_c RoIs per query
    ap.add_argument('--cpu-iters', type=int, default=24)       # ~10 s of CPU work on the bounded sample
    ap.add_argument('--cpu-threads', type=int, default=16)
    ap.add_argument('--cpu-timeout', type=int, default=150)
    args = ap.parse_args()

    from mv2d_amd import dist as mdist
    from mv2d_amd import ops, synthetic
    from mv2d_amd.engine import HeadEngine
    import torch.distributed as dist

    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
    local = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)    # one rank per GPU
    torch.cuda.set_device(local)
    rank, world, _ = mdist.init_from_env(backend=args.backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    collective = world > 1 or os.environ.get('MV2D_FORCE_COLLECTIVE', '0') == '1'    # the per-step all-gather of decoded boxes
    dev = torch.device('cuda', local)

    prob = synthetic.make_problem(args.workload, seed=rank)        # weak scaling: every rank its own frames
    kind = prob['kind']
    sd = synthetic.make_head_state(seed=0)
    base = HeadEngine(sd, kind, dev, num_views=prob['views_per_frame'], topk=args.corr_topk)
    base.force_nc = args.force_nc
    base.fork_qg = args.inflight == 1
    engines = [base] + [base.clone_shared() for _ in range(args.inflight - 1)]
    # frames in flight go on streams that were MEASURED to run concurrently (queue/pipe sharing serialises others)
    from mv2d_amd.streams import concurrent_streams
    if os.environ.get('MV2D_NAIVE_STREAMS', '0') == '1':
        streams = [torch.cuda.Stream(device=dev) for _ in range(args.inflight)]
    else:
        pool = concurrent_streams(min(args.inflight, 4), dev)
        streams = [pool[i % len(pool)] for i in range(args.inflight)]
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = prob['img_metas']
    B = args.batch
    if B > 1:                                                      # the samples of a batch are different frames
        more = [synthetic.make_problem(args.workload, seed=1000 * (b + 1) + rank) for b in range(B - 1)]
        # the producer (backbone + neck of a batch) hands over one stacked [B*V,256,h,w] map
        feats_b = torch.cat([feat] + [torch.from_numpy(m['feat']).to(dev) for m in more], 0).contiguous()
        props_b = [props] + [[torch.from_numpy(p) for p in m['proposals']] for m in more]
        metas_b = [metas] + [m['img_metas'] for m in more]
    use_graph = not args.no_graph
    # ping-pong payload buffers: the streams free-run (no per-step join on one GPU); with N > 1 the all-gather of step k
    # runs on the main stream behind the frames of step k while the frames of step k+1 are already executing.
    payload = [torch.zeros((args.inflight * B, 300 * 11 + 1), device=dev) for _ in range(2)]
    gathered_ev = [None, None]
    step_no = [0]

    def step():
        k = step_no[0] & 1
        step_no[0] += 1
        cur = torch.cuda.current_stream()
        done = []
        for i, (e, s) in enumerate(zip(engines, streams)):
            with torch.cuda.stream(s):
                if gathered_ev[k] is not None:
                    s.wait_event(gathered_ev[k])
                if B > 1:
                    o = e.run_batch(feats_b, props_b, metas_b, use_graph=use_graph)
                else:
                    o = e.run(feat, props, metas, use_graph=use_graph)
                ops.pack_detections(o['boxes'], o['scores'], o['labels'], o['count'], payload[k][i * B:(i + 1) * B])
                if collective:
                    ev = torch.cuda.Event()
                    ev.record()
                    done.append(ev)
        if collective:
            for ev in done:
                cur.wait_event(ev)
            out = mdist.gather_detections(payload[k])     # the one collective of an evaluation step (RCCL all-gather)
            gathered_ev[k] = torch.cuda.Event()
            gathered_ev[k].record()
            return out
        return payload[k]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    samples = world * args.inflight * B * args.steps
    value = samples / elapsed

    # ---------------- per-stage timing of the same frame with HIP events on the launch stream (single stream, eager)
    eng = base
    run_once = (lambda: eng.run_batch(feats_b, props_b, metas_b)) if B > 1 else (lambda: eng.run(feat, props, metas))
    out0 = run_once()                                    # the launches of the timed region: B samples each
    torch.cuda.synchronize()
    R = out0['R']
    ws = out0['ws']
    S = int(ws['S_dev'].item())
    nnz = int(ws['nnz'][0].item())
    for _ in range(3):
        run_once()                                       # eager warm-up of the instrumented path
    torch.cuda.synchronize()
    eng.prof = {}
    n_prof = max(5, min(args.steps, 20))
    for _ in range(n_prof):
        run_once()
    torch.cuda.synchronize()
    prof, eng.prof = eng.prof, None
    names = list(prof.keys())
    stage_ms = {}
    for a, b in zip(names[:-1], names[1:]):
        stage_ms[a] = statistics.median(x.elapsed_time(y) for x, y in zip(prof[a], prof[b]))
    fl, by = stage_flops(kind, R, S), stage_bytes(kind, R, S)
    dom = max(fl, key=lambda k: stage_ms.get(k, 0.0))
    dom_ms = stage_ms[dom]
    tflops = fl[dom] / (dom_ms * 1e-3) / 1e12
    gbs = by[dom] / (dom_ms * 1e-3) / 1e9
    # HBM traffic of that launch from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected offline with
    # tools/rocpd_pmc.py and committed under profiles/): counters cannot be read from inside this process
    traffic, traffic_detail = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        t = pmc.get(f'{args.workload}@{B}', {}).get(dom)              # counters of a launch with the same number of samples
        if t:
            traffic = t['fetch_bytes'] + t['write_bytes']            # HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)
            traffic_detail = {'fetch_bytes': t['fetch_bytes'], 'write_bytes': t['write_bytes'],
                              'source': 'profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'}
    except Exception:
        pass
    # the roofline that binds this launch is the one it sits closer to
    f_mfma, f_hbm = tflops / PEAK_BF16_TFLOPS, gbs / PEAK_HBM_GBS
    kname = f"{STAGE_KERNEL.get(dom, 'gemm_bf16_kernel')}[{dom}]"
    if f_hbm > f_mfma:
        roofline = dict(bound='hbm', kernel=kname, achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(f_hbm, 4),
                        traffic=traffic, launch_ms=round(dom_ms, 4), bytes_per_launch=by[dom], flops_per_launch=fl[dom],
                        mfma_frac=round(f_mfma, 4))
    else:
        roofline = dict(bound='mfma', kernel=kname, achieved=round(tflops, 2), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s',
                        frac=round(f_mfma, 4), traffic=traffic, launch_ms=round(dom_ms, 4), flops_per_launch=fl[dom],
                        bytes_per_launch=by[dom], hbm_frac=round(f_hbm, 4))
    roofline['traffic_detail'] = traffic_detail
    stage_roofline = {k: dict(ms=round(stage_ms[k], 4), tflops=round(fl[k] / (stage_ms[k] * 1e-3) / 1e12, 1),
                              gbs=round(by[k] / (stage_ms[k] * 1e-3) / 1e9, 1)) for k in fl if k in stage_ms}

    # ---------------- decoder ms/iter (CrossAttentionBoxHead transformer on prepared inputs), hipGraph replay
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng._enqueue_decoder(ws, R)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        eng._enqueue_decoder(ws, R)
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    decoder_ms = e0.elapsed_time(e1) / 50

    # ---------------- CPU baseline: the oracle (port of the reference algorithm) on the host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import subprocess
        threads = min(os.cpu_count() or 1, args.cpu_threads)
        try:
            r = subprocess.run([sys.executable, '-m', 'oracle.cpu_baseline', '--workload', args.workload, '--iters', str(args.cpu_iters),
                                '--threads', str(threads)], cwd=ROOT, capture_output=True, text=True, timeout=args.cpu_timeout)
            lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
            cpu = json.loads(lines[-1]) if lines else dict(value=None, unit='samples/s', cores=threads, kind='port',
                                                           sample=f'oracle subprocess failed: {r.stderr[-200:]}')
        except subprocess.TimeoutExpired:
            cpu = dict(value=None, unit='samples/s', cores=threads, kind='port', sample=f'oracle did not finish {args.cpu_iters}+1 frames in {args.cpu_timeout}s')

    if rank == 0:
        line = {
            'metric': 'multi-view samples/sec (6-cam frames) through the MV2D RoI-head hot path',
            'value': round(value, 2), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16 (key side MFMA) / f32 + bf16x3 split precision (query side)', 'data': 'synthetic',
            'config': {'workload': f'{args.workload}: MV2D-{kind} head, {len(metas)} views {metas[0]["img_shape"][1]}x{metas[0]["img_shape"][0]}, '
                                   f'R={R} queries, S={S} key positions, nnz={nnz} allowed (q,k) pairs' + (f' (totals of the {B} samples of a launch)' if B > 1 else '') + (f', corr_topk={args.corr_topk}' if args.corr_topk else '') + (f', forced n_c={args.force_nc}' if args.force_nc else ''),
                       'frames_per_step_per_gpu': args.inflight * B, 'global_batch': world * args.inflight * B,
                       'streams_per_gpu': args.inflight, 'samples_per_launch': B, **({'pe_sine_branch': 'per-geometry table (opt-in MV2D_PE_SINE_TABLE=1, not the default path)'} if SINE_TABLE else {}),
                       'parallelism': f'dp{world}', 'hipgraph': use_graph},
            'decoder_ms_per_iter': round(decoder_ms / B, 4), 'decoder_ms_per_launch': round(decoder_ms, 4),
            'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
            'roofline': roofline, 'stage_roofline': stage_roofline,
            'cpu_baseline': cpu,
        }
    # RCCL writes its version banner through C stdio (block-buffered on a pipe, so it would come out at process exit, after the JSON
    # line): every rank flushes it now, the barrier orders that before rank 0 prints, and the JSON line stays the last line of stdout
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
