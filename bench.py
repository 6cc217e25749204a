#!/usr/bin/env python
"""bench.py — MV2D RoI-head hot path on MI355X: multi-view samples/s + decoder ms/iter (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            # no launcher: re-executes itself under torch.distributed.run, one rank per GPU

The timed loop runs the engine's route: INDEX-EXACT (fp16 hi + lo pairs on both sides; `route` in the line); `--key16` times the opt-in mode with one
fp16 rounding of the key side instead, and the default line carries that mode as a labelled extra leg.

One "step" = one pass of the whole hot path (PE -> RoI gather -> query generator -> box correlation -> key list/CSR
-> 6-layer decoder with the tile cross attention -> heads -> top-k decode, + the all-gather of decoded boxes when N > 1) over
`--inflight` x `--batch` synthetic frames per GPU: `--inflight` HIP streams, each replaying hipGraphs whose launches carry `--batch`
samples.  Every stream has its OWN frames and rotates through `--rotate` distinct frame sets (feature maps, 2-D boxes) from step to
step; on the two-frame (T) workloads every sample of every step also has its own img_metas (ego motion, time stamps), so the
calibration tables are rebuilt on the host and uploaded inside the timed loop.  Inputs are resident in HBM before the timed region.
Extra legs after the timed region (same process, reported beside `value`): the round-1 protocol (every stream replays the same
frames), one sample per launch on 4 streams, and one sample on one stream with a synchronisation per frame (latency).
Prints ONE JSON line.
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



def stage_flops(kind, R, S, L=6):
    """Algorithmic FLOPs (2 x MAC) of the dense bf16 MFMA launches of one frame."""
    # (the input-independent sine branch of the PE block is folded into a per-(weights, geometry) table, LOG.md section 8: its FLOPs are NOT counted)
    return {
        'pe_fused': 2.0 * S * (192 * 1024 + 1024 * 256 + 2 * 256 * 256),                       # 1.16 MFLOP per key position
        'qg_conv_gemm': 2.0 * R * 49 * 2304 * 256,
    }


def stage_bytes(kind, R, S, L=6):
    """Algorithmic HBM bytes of the same launches: A read once + weights once + output written once (bf16 = 2 B, fp32 = 4 B)."""
    return {
        # A1 + Xf key16 (fp16) in, table row fp32 in, the four weight matrices once; S path: pe fp32 out (RoIAlign reads it; its keys are
        # RoI-aligned rows); T path: fp32 feature row in, Xk key16 out (nothing reads pe there)
        'pe_fused': (S * (192 + 256) * 2 + S * 256 * 4 + (192 * 1024 + 1024 * 256 + 2 * 256 * 256) * 2 +
                     (S * 256 * (4 + 2) if kind == 'T' else S * 256 * 4)),
        'qg_conv_gemm': R * 49 * 256 * 2 + 2304 * 256 * 2 + R * 256 * 4,                       # pooled [R,256] output
    }


STAGE_KERNEL = {'pe_fused': 'pe_tab_kernel (frustum MLP + gate, sine branch from the table)', 'qg_conv_gemm': 'roi_conv_pool_kernel (conv3x3 + ReLU + avgpool fused)'}


def build_step(engs, strs, sets, pool, Bs, pay, state, *, collective, use_graph, rotate=True, pack=None, gather=None, cuda=True, rounds=1):
    """One bench step over `len(engs)` streams: every stream runs its engine ``rounds`` times, each time on its next frame set (``Bs`` samples per
    launch), packs the decoded boxes into rows [(i * rounds + g) * Bs, + Bs) of the ping-pong payload buffer ``pay[k]``; with ``collective`` the main stream then
    waits for all of them and runs the ONE all-gather of the step (mv2d_amd.dist.gather_detections) while the streams already work on the
    next step's frames in the other buffer.  ``state`` = dict(gathered_ev=[None, None], step_no=0) shared by the steps built on the same
    buffers.  ``cuda=False`` (tests/test_dist_gloo_cpu.py, world-2 gloo): the same control flow without HIP streams / events, ``pack`` /
    ``gather`` then default to the torch formulations of mv2d_amd.dist."""
    import contextlib
    from mv2d_amd import dist as mdist
    if pack is None:
        if cuda:
            from mv2d_amd import ops
            pack = ops.pack_detections
        else:
            pack = lambda bx, sc, lb, cnt, out: out.copy_(mdist.pack_detections_batch(bx.view(-1, *bx.shape[-2:]), sc.view(-1, sc.shape[-1]),  # noqa: E731
                                                                                       lb.view(-1, lb.shape[-1]), cnt.view(-1)))
    gather = gather or mdist.gather_detections
    cnt = [0]

    def step():
        n = cnt[0]
        cnt[0] += 1
        k = state['step_no'] & 1
        state['step_no'] += 1
        cur = torch.cuda.current_stream() if cuda else None
        done = []
        # round-major: every pass visits every stream once, so the host's wait for an engine's previous frame (its pinned staging buffers) falls behind
        # the launches of the other streams, as in the one-round step
        for g_ in range(rounds):
            for i, (e, s_) in enumerate(zip(engs, strs)):
                with (torch.cuda.stream(s_) if cuda else contextlib.nullcontext()):
                    if g_ == 0 and cuda and state['gathered_ev'][k] is not None:
                        s_.wait_event(state['gathered_ev'][k])
                    n_ = n * rounds + g_
                    fb, pb, mb = sets[i][(n_ % len(sets[i])) if rotate else 0]
                    if pool is not None and rotate:
                        mb = [pool[i][(n_ * Bs + b) % len(pool[i])] for b in range(Bs)]
                    dst = pay[k][(i * rounds + g_) * Bs:(i * rounds + g_ + 1) * Bs]
                    if cuda:
                        # the decode kernel writes the sample's wire rows itself (one launch less per frame than mv2d_pack_detections)
                        o = e.run_batch(fb, pb, mb, use_graph=use_graph, payload=dst) if Bs > 1 else e.run(fb, pb[0], mb[0], use_graph=use_graph, payload=dst)
                    else:
                        o = e.run_batch(fb, pb, mb, use_graph=use_graph) if Bs > 1 else e.run(fb, pb[0], mb[0], use_graph=use_graph)
                        pack(o['boxes'], o['scores'], o['labels'], o['count'], dst)
                    if g_ == rounds - 1 and collective and cuda:
                        ev = torch.cuda.Event()
                        ev.record()
                        done.append(ev)
        if collective:
            for ev in done:
                cur.wait_event(ev)
            out = gather(pay[k])     # the one collective of an evaluation step (RCCL all-gather; gloo in the CPU test)
            if cuda:
                state['gathered_ev'][k] = torch.cuda.Event()
                state['gathered_ev'][k].record()
            return out
        return pay[k]
    return step


class StubEngine:
    """--stub-engine ONLY (tests/test_dist_gloo_cpu.py: the N > 1 control flow of this script end to end on CPU ranks): stands for HeadEngine in
    build_step with deterministic "decoded boxes" per (rank, stream, call).  No kernel runs; a line produced with it says so and is not a measurement."""

    def __init__(self, rank, stream):
        self.rank, self.stream, self.calls = rank, stream, 0

    def run_batch(self, fb, pb, mb, use_graph=False):
        B = len(pb)
        n = torch.tensor([(self.rank * 7 + self.stream * 3 + self.calls + b) % 300 + 1 for b in range(B)], dtype=torch.int32)
        boxes = torch.zeros(B, 300, 9); scores = torch.zeros(B, 300); labels = torch.zeros(B, 300, dtype=torch.int64)
        for b in range(B):
            boxes[b, :n[b]] = 100.0 * self.rank + 10.0 * self.stream + self.calls + 0.1 * b
            scores[b, :n[b]] = 0.25
            labels[b, :n[b]] = (self.calls + b) % 10
        self.calls += 1
        return dict(boxes=boxes, scores=scores, labels=labels, count=n)

    def run(self, fb, p0, m0, use_graph=False):
        return self.run_batch(fb, [p0], [m0], use_graph)


def stub_main(args):
    """The timed loop, the barriers, the max over ranks, the per-step all-gather and the JSON line of main() with StubEngine instead of the HIP engine
    (no GPU): what `python bench.py --gpus N --backend gloo --stub-engine` runs, self-spawn included."""
    import torch.distributed as dist
    from mv2d_amd import dist as mdist
    rank, world, _ = mdist.init_from_env(backend=args.backend or 'gloo')
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    B, G, NI = args.batch, max(1, args.rounds), args.inflight
    engs = [StubEngine(rank, i) for i in range(NI)]
    sets = [[(None, [None] * B, [None] * B)] for _ in range(NI)]
    pay = [torch.zeros((NI * G * B, 300 * 11 + 1)) for _ in range(2)]
    state = dict(gathered_ev=[None, None], step_no=0)
    collective = world > 1 or os.environ.get('MV2D_FORCE_COLLECTIVE', '0') == '1'
    step = build_step(engs, [None] * NI, sets, None, B, pay, state, collective=collective, use_graph=False, cuda=False, rounds=G)

    def barrier():
        if world > 1:
            dist.barrier()
    for _ in range(args.prime + args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_last = (state['step_no'] - 1) & 1
    check = None
    if collective:
        digest = torch.tensor([float(out.double().sum())], dtype=torch.float64)
        digests = [torch.zeros_like(digest) for _ in range(world)]
        if world > 1:
            dist.all_gather(digests, digest)
        else:
            digests = [digest]
        check = dict(backend=dist.get_backend() if dist.is_initialized() else None, world=world, gathered_shape=list(out.shape),
                     gathered_equals_packed=bool(torch.equal(out[rank], pay[k_last])), payload_nonzero_entries=int((pay[k_last] != 0).sum()),
                     every_rank_holds_the_same_gathered_tensor=bool(all(float(d) == float(digests[0]) for d in digests)),
                     steps_with_collective=state['step_no'])
    samples = world * NI * G * B * args.steps
    line = dict(metric='multi-view samples/sec -- STUB ENGINE: control flow only, no kernel ran, not a measurement', stub_engine=True,
                value=round(samples / elapsed, 2), unit='samples/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                ms_per_step=round(elapsed / args.steps * 1e3, 4), higher_is_better=True, scaling='weak', vs_baseline=None, data='stub',
                timed_seconds=round(elapsed, 6),
                config=dict(workload='stub', frames_per_step_per_gpu=NI * G * B, global_batch=world * NI * G * B, streams_per_gpu=NI,
                            launch_sequences_per_stream_and_step=G, samples_per_launch=B, parallelism=f'dp{world}'),
                collective_check=check)
    sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


def golden_for(args):
    """The reference's own output on the seed-0 frame of the workload (tests/golden/<workload>.npz, oracle/gen_golden.py), or None."""
    gpath = os.path.join(ROOT, 'tests', 'golden', args.workload + '.npz')
    if os.path.exists(gpath) and args.corr_topk is None and args.force_nc is None:
        return np.load(gpath)
    return None


def index_mismatches(e, gold, feat, props, metas):
    """Integer parity of one engine run against the reference golden: ranked labels / ranked (query, class) indices / top-k set entries that
    differ, and the largest score difference."""
    o_ = e.run(feat, props, metas)
    torch.cuda.synchronize()
    n = int(o_['count'].item())
    gl, ref = gold['labels'], gold['topk_index']
    m = min(n, len(gl))
    lab = o_['labels'][:n].cpu().numpy()
    flat = o_['bbox_index'][:n].cpu().numpy() * 10 + lab
    d = dict(ranked_labels=int((lab[:m] != gl[:m]).sum()) + abs(n - len(gl)), of=int(len(gl)))
    if len(ref) == len(gl):
        d['ranked_indices'] = int((flat[:m] != ref[:m]).sum()) + abs(n - len(ref))
        d['topk_set'] = len(set(flat.tolist()) - set(ref.tolist()))
    d['max_score_err'] = float(np.abs(o_['scores'][:m].cpu().numpy() - gold['scores'][:m]).max())
    return d


def ws_rows(out):
    """rows the launches of a frame run on (the RoI-count bucket, >= the real R)."""
    return int(out['ws']['x'].shape[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)       # 60 steps x ~50 ms: a timed region of ~3 s
    ap.add_argument('--rounds', type=int, default=6, help='launch sequences per stream and step (a step = inflight x rounds x batch frames per GPU): 6 makes the 20 '
                                                          'steps of the driver a timed region of ~1 s instead of 0.17 s (round 6; 1 = the step of rounds 1-5)')
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--prime', type=int, default=40, help='untimed setup steps in front of the warm-up (graph captures, clock ramp)')
    ap.add_argument('--workload', default='cfg2_s', help='cfg2_s (MV2D-S 6 cams 1408x512, headline) | cfg3_t | cfg5_t | cfg1_s ...')
    ap.add_argument('--inflight', type=int, default=4, help='HIP streams per GPU, each running its own launch sequence per step')
    ap.add_argument('--batch', type=int, default=16, help='samples sharing every launch of a stream (HeadEngine.run_batch); 16 since round 3 (8: -5 %%)')
    ap.add_argument('--rotate', type=int, default=4, help='distinct frame sets every stream cycles through (1: the same frames every step)')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip the fixed-input / batch-1 / latency legs')
    ap.add_argument('--nchw-input', action='store_true', help='feature maps in contiguous NCHW memory (rounds 1-5: the engine transposes the rows it reads) instead of '
                                                             'channels_last = position-major, the layout mv2d_amd.plugin.neck emits (no transposition; round 6 default)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL; gloo only for single-GPU dry runs)')
    ap.add_argument('--corr-topk', type=int, default=None)    # S path: correlated RoIs per other view (reference default 1); T path: 20
    ap.add_argument('--force-nc', type=int, default=None)     # S path sweep: synthetic correlation lists, n_c RoIs per query
    ap.add_argument('--cpu-iters', type=int, default=10)       # timed frames of the CPU baseline (3 warm-ups + 10 timed)
    ap.add_argument('--cpu-threads', type=int, default=0)       # 0: sweep 8 / 16 / 32 / 64 threads first and use the best; > 0: that count
    ap.add_argument('--cpu-timeout', type=int, default=150)
    ap.add_argument('--cpu-all-budget', type=int, default=15, help='seconds of the bounded all-cores CPU leg (BASELINE.md protocol: os.cpu_count() threads, 3 warm-ups)')
    ap.add_argument('--key16', action='store_true', help='run the timed loop in the OPT-IN key16 mode (HeadEngine(exact=False): one fp16 rounding of the key side, '
                                                        'ranked indices differ from the reference) instead of the index-exact route, which is the default since round 5')
    ap.add_argument('--exact', action='store_true', help='(round-4 flag; the index-exact route is the default now -- accepted and ignored)')
    ap.add_argument('--stub-engine', action='store_true', help='plumbing test on CPU ranks (gloo): the whole step / barrier / all-gather / JSON flow with a stub in place of the HIP engine; NOT a measurement')
    ap.add_argument('--spawn-check', action='store_true', help='only initialise the process group, all-gather the ranks, print one JSON line (no GPU work): the self-spawn test')
    ap.add_argument('--brief', action='store_true', help='headline timing + tile-kernel roofline only (what the other_workloads legs of the default run call)')
    ap.add_argument('--no-other-workloads', action='store_true', help='skip the short cfg3_t / cfg5_t legs (sub-processes of this script)')
    ap.add_argument('--no-parity-leg', action='store_true', help='skip the single-sample run that counts the integer mismatches against the reference golden')
    ap.add_argument('--min-seconds', type=float, default=1.0, help='when the K timed steps take less than this, a second, longer loop of the same step is timed and reported beside them')
    ap.add_argument('--xattn-waves', type=int, default=None, help='waves per query of the tile cross-attention kernel (engine default: 2)')
    ap.add_argument('--fuse-maps', type=int, default=None, help='1 / 0: force the per-head maps of the tile attention into / out of the neighbouring row kernels (engine default: by row count)')
    ap.add_argument('--fuse-xattn', type=int, default=None, help='1 / 0: force the one-launch cross attention (csrc/xattn_fused.hip) on / off (engine default: on for the S path)')
    ap.add_argument('--force-collective', action='store_true', help='one rank: initialise the process group (nccl = RCCL) anyway and run the per-step all-gather of decoded boxes')
    ap.add_argument('--no-collective-leg', action='store_true', help='skip the one-rank RCCL leg (a sub-process of this script with --force-collective)')
    args = ap.parse_args()
    if args.force_collective:
        os.environ['MV2D_FORCE_COLLECTIVE'] = '1'
        if 'MASTER_PORT' not in os.environ:
            import socket
            s_ = socket.socket()
            s_.bind(('127.0.0.1', 0))
            os.environ['MASTER_PORT'] = str(s_.getsockname()[1])
            s_.close()
    if args.brief:
        args.no_extra_legs = args.no_cpu_baseline = args.no_other_workloads = args.no_collective_leg = True
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (tools/dist_test.sh:10-22 of the
        # reference does the same with torch.distributed.launch).  The driver's own `python -m torch.distributed.run ... bench.py --gpus N` sets
        # WORLD_SIZE and never comes here.
        import socket
        s_ = socket.socket()
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if args.stub_engine:
        return stub_main(args)
    if args.spawn_check:
        import torch.distributed as dist_
        from mv2d_amd import dist as mdist_
        rank_, world_, _ = mdist_.init_from_env(backend=args.backend)
        assert world_ == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world_}'
        got = [None] * world_
        if world_ > 1:
            dist_.all_gather_object(got, rank_)
            dist_.barrier()
            dist_.destroy_process_group()
        else:
            got = [0]
        if rank_ == 0:
            print(json.dumps(dict(spawn_check=True, world=world_, ranks=got, backend=args.backend or 'nccl')), flush=True)
        return

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
    base = HeadEngine(sd, kind, dev, num_views=prob['views_per_frame'], topk=args.corr_topk, exact=not args.key16)
    base.force_nc = args.force_nc
    if args.xattn_waves:
        base.xattn_waves = args.xattn_waves
    if args.fuse_maps is not None:
        base.fuse_maps = bool(args.fuse_maps)
    if args.fuse_xattn is not None:
        base.fuse_xattn = bool(args.fuse_xattn)
    base.fork_qg = args.inflight == 1
    engines = [base] + [base.clone_shared() for _ in range(args.inflight - 1)]
    # frames in flight go on streams that were MEASURED to run concurrently (queue/pipe sharing serialises others)
    from mv2d_amd.streams import concurrent_streams
    pool = concurrent_streams(min(args.inflight, 4), dev)
    streams = [pool[i % len(pool)] for i in range(args.inflight)]
    feat = torch.from_numpy(prob['feat']).to(dev)
    props = [torch.from_numpy(p) for p in prob['proposals']]
    metas = prob['img_metas']
    B = args.batch
    use_graph = not args.no_graph
    two_frame = prob['frames'] > 1
    K = max(1, args.rotate)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)

    CL = torch.channels_last

    def frame_sets(n_streams, Bs, k_sets, vary_rois=False, nchw=None):
        nchw = args.nchw_input if nchw is None else nchw
        """[stream][set] -> (feats [Bs*V,256,h,w] on the device, proposals per sample, metas per sample): every stream its own frames.
        Set 0 of stream 0 starts with the seed-0 problem (the one the CPU baseline and the stage timings use)."""
        out = []
        for i in range(n_streams):
            sets = []
            for k in range(k_sets):
                fl, pl, ml = [], [], []
                for b in range(Bs):
                    if i == 0 and k == 0 and b == 0:
                        fl.append(feat); pl.append(props); ml.append(metas)
                        continue
                    seed_ = 100000 * (i + 1) + 1000 * k + 10 * b + rank
                    m = synthetic.make_problem(args.workload, seed=seed_, with_feat=False)
                    if vary_rois:
                        # every sample its own number of 2-D boxes: 5/6 .. 7/6 of the nominal count per view (250 .. 350 per 300-query sample)
                        g_ = np.random.Generator(np.random.PCG64(seed_))
                        wl = synthetic.WORKLOADS[args.workload]
                        n_nom = wl[6]
                        counts_ = [int(x) for x in g_.integers(n_nom * 5 // 6, n_nom * 7 // 6 + 1, len(m['proposals']))]
                        m['proposals'] = synthetic.make_proposals(len(counts_), counts_, wl[3], wl[4], seed_ + 1)
                    fl.append(torch.randn(feat.shape, device=dev, generator=gen))
                    pl.append([torch.from_numpy(p) for p in m['proposals']]); ml.append(m['img_metas'])
                fm = torch.cat(fl, 0) if Bs > 1 else fl[0]
                # (logical shape [V,256,h,w] either way; channels_last memory = one 1 KB row per map cell, what the neck of this package writes)
                sets.append((fm.contiguous() if nchw else fm.contiguous(memory_format=CL), pl, ml))
            out.append(sets)
        return out

    def meta_pool(n_streams, n):
        """Two-frame workloads: n distinct per-sample img_metas per stream (own ego motion / time stamps each).  n exceeds the engine's
        table cache (64 entries, FIFO), so every use rebuilds the camera tables on the host -- as a stream of real frames would."""
        if not two_frame:
            return None
        return [[synthetic.make_problem(args.workload, seed=0, with_feat=False, ego=0.003 * (j + 1) + 0.5 * i)['img_metas'] for j in range(n)]
                for i in range(n_streams)]

    sets_main = frame_sets(args.inflight, B, K)
    pool_main = meta_pool(args.inflight, 96)
    feats_b, props_b, metas_b = sets_main[0][0]                    # stage timings / decoder leg: the first set of stream 0
    # ping-pong payload buffers: the streams free-run (no per-step join on one GPU); with N > 1 the all-gather of step k
    # runs on the main stream behind the frames of step k while the frames of step k+1 are already executing.
    G = max(1, args.rounds)
    payload = [torch.zeros((args.inflight * G * B, 300 * 11 + 1), device=dev) for _ in range(2)]
    state = dict(gathered_ev=[None, None], step_no=0)

    def make_step(engs, strs, sets, pool, Bs, pay, rotate=True, rounds=1):
        return build_step(engs, strs, sets, pool, Bs, pay, state, collective=collective, use_graph=use_graph, rotate=rotate, rounds=rounds)

    step = make_step(engines, streams, sets_main, pool_main, B, payload, rounds=G)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.prime):                     # setup (graph captures, clocks), not part of the W warm-up steps
        step()
    if args.prime:
        barrier()
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
    samples = world * args.inflight * G * B * args.steps
    value = samples / elapsed
    # the contract's K steps may be a very short region (20 steps = 0.08 s): time a second, longer loop of the same step so that an
    # external sampler sees the GPU busy; both are reported, `value` stays the K-step number
    long_run = None
    if elapsed < 0.9 * args.min_seconds and not args.brief:
        n_long = int(min(5000, max(args.steps, args.min_seconds * 1.2 / (elapsed / args.steps))))
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_long):
            step()
        barrier()
        el_long = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el_long], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_long = float(t.item())
        long_run = dict(steps=n_long, seconds=round(el_long, 3), samples_s=round(world * args.inflight * G * B * n_long / el_long, 2))

    # ---------------- with a collective in the step: the gathered tensor's slice of this rank must be bit-identical to the payload it packed
    collective_check = None
    if collective:
        out_g = step()
        torch.cuda.synchronize()
        k_last = (state['step_no'] - 1) & 1
        mine = out_g[rank].contiguous().view(torch.int32)
        collective_check = dict(backend=dist.get_backend(), world=world, gathered_shape=list(out_g.shape),
                                gathered_equals_packed=bool(torch.equal(mine, payload[k_last].view(torch.int32))),
                                payload_nonzero_entries=int((payload[k_last] != 0).sum().item()),
                                steps_with_collective=state['step_no'], hipgraph=use_graph)

    # ---------------- extra legs (single GPU): what the headline's batching / streams / input rotation are worth
    extra = dict(samples_s_rotating_inputs=round(value, 2), rotating_frame_sets_per_stream=K,
                 img_metas='unique per sample and step (camera tables rebuilt + uploaded in the timed loop)' if two_frame else
                           'one static rig (single-frame workload): tables uploaded once')
    if world == 1 and not args.no_extra_legs:
        def timed(fn, n_steps, n_warm):
            for _ in range(n_warm):
                fn()
            barrier()
            t_ = time.perf_counter()
            for _ in range(n_steps):
                fn()
            barrier()
            return time.perf_counter() - t_
        n_x = max(10, min(args.steps * G, 100))
        # (a) the round-1 protocol: every stream replays the same frames, img_metas never change
        same = [[sets_main[0][0]] for _ in range(args.inflight)]
        el = timed(make_step(engines, streams, same, None, B, payload, rotate=False), n_x, 3)
        extra['samples_s_fixed_inputs'] = round(args.inflight * B * n_x / el, 2)
        # (a0) the other input layout (contiguous NCHW unless --nchw-input: then channels_last)
        sets_o = frame_sets(args.inflight, B, K, nchw=not args.nchw_input)
        el = timed(make_step(engines, streams, sets_o, pool_main, B, payload), n_x, K + 1)
        extra['samples_s_channels_last_input' if args.nchw_input else 'samples_s_nchw_input'] = round(args.inflight * B * n_x / el, 2)
        del sets_o
        # (a1) round-over-round continuity (the default moved from 8 to 16 samples per launch at the end of round 3): the same protocol at 8
        if B == 16:
            sets8 = frame_sets(args.inflight, 8, K)
            pay8 = [torch.zeros((args.inflight * 8, 300 * 11 + 1), device=dev) for _ in range(2)]
            el = timed(make_step(engines, streams, sets8, pool_main, 8, pay8), n_x, K + 1)
            extra['samples_s_batch8'] = round(args.inflight * 8 * n_x / el, 2)
            del sets8, pay8
        # (a2) the number of RoIs differs from sample to sample and from step to step (launches run on 64-row buckets, one graph per bucket)
        sets_v = frame_sets(args.inflight, B, K, vary_rois=True)
        el = timed(make_step(engines, streams, sets_v, pool_main, B, payload), n_x, K + 1)
        extra['samples_s_varying_rois'] = round(args.inflight * B * n_x / el, 2)
        extra['varying_rois_per_sample'] = [int(sum(p.shape[0] for p in pl_)) for pl_ in sets_v[0][0][1]][:8]
        del sets_v
        # (a3) OPT-IN HeadEngine.last_stage_heads: the cls / reg branches of the last decoder layer only (inference reads nothing else; the
        # reference's forward evaluates all six, so the headline does too)
        for e in engines:
            e.last_stage_heads = True                    # (part of the graph key: the warm-up steps capture the other graphs)
        el = timed(make_step(engines, streams, sets_main, pool_main, B, payload), n_x, K + 1)
        extra['samples_s_last_stage_heads_only'] = round(args.inflight * B * n_x / el, 2)
        for e in engines:
            e.last_stage_heads = False
        # (b) one sample per launch (the reference's call shape) on the same streams, rotating inputs
        sets1 = frame_sets(args.inflight, 1, K) if B > 1 else sets_main
        pool1 = meta_pool(args.inflight, 96) if B > 1 else pool_main
        pay1 = [torch.zeros((args.inflight, 300 * 11 + 1), device=dev) for _ in range(2)]
        el = timed(make_step(engines, streams, sets1, pool1, 1, pay1), n_x, K + 1)
        extra['samples_s_batch1'] = round(args.inflight * n_x / el, 2)
        # (c) one stream, one sample, host synchronisation after every frame: the latency a single rig sees
        lat = []
        for n_ in range(K + 1 + n_x):
            fb, pb, mb = sets1[0][n_ % len(sets1[0])]
            m0 = pool1[0][n_ % len(pool1[0])] if pool1 is not None else mb[0]
            t_ = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                o_ = engines[0].run(fb, pb[0], m0, use_graph=use_graph, payload=pay1[0][:1])
            streams[0].synchronize()
            if n_ > K:
                lat.append(time.perf_counter() - t_)
        extra['latency_ms_single_stream'] = round(statistics.median(lat) * 1e3, 4)
        extra['samples_s_single_stream'] = round(1.0 / statistics.median(lat), 2)
        # (d) the OTHER mode (the headline runs the index-exact route unless --key16: then this leg is the index-exact one), same protocol as the
        # headline: same streams, same rotating frame sets, same batch.  key16 = HeadEngine(exact=False): ONE fp16 rounding of the key side, opt-in.
        alt_base = HeadEngine(sd, kind, dev, num_views=prob['views_per_frame'], topk=args.corr_topk, exact=args.key16)
        alt_base.fork_qg = False
        alt_engines = [alt_base] + [alt_base.clone_shared() for _ in range(args.inflight - 1)]
        n_e = max(10, min(args.steps * G, 60))
        el = timed(make_step(alt_engines, streams, sets_main, pool_main, B, payload), n_e, K + 1)
        alt_v = round(args.inflight * B * n_e / el, 2)
        ex_v, k16_v = (alt_v, value) if args.key16 else (value, alt_v)
        extra['samples_s_index_exact'] = round(ex_v, 2)
        extra['samples_s_key16_mode_opt_in'] = round(k16_v, 2)
        extra['index_exact_vs_key16_mode'] = round(ex_v / k16_v, 3)
        # (e) integer parity of both modes against the REFERENCE's own output on the seed-0 frame of this workload (tests/golden/<workload>.npz,
        # produced by the unmodified reference, oracle/gen_golden.py): ranked labels / ranked (query, class) indices / top-k set entries that differ
        gold = golden_for(args)
        if gold is not None:
            ex_e, k16_e = (alt_base, base) if args.key16 else (base, alt_base)
            extra['index_mismatches'] = dict(index_exact=index_mismatches(ex_e, gold, feat, props, metas), key16_mode=index_mismatches(k16_e, gold, feat, props, metas),
                                             reference='tests/golden/%s.npz (unmodified reference, seed-0 frame)' % args.workload)
        del alt_engines, alt_base

    if 'index_mismatches' not in extra and world == 1 and not args.no_parity_leg:
        gold = golden_for(args)
        if gold is not None:            # (--brief / --no-extra-legs: the route of the timed loop only)
            extra['index_mismatches'] = {'key16_mode' if args.key16 else 'index_exact': index_mismatches(base, gold, feat, props, metas),
                                         'reference': 'tests/golden/%s.npz (unmodified reference, seed-0 frame)' % args.workload}
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
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))).get(f'{args.workload}@{B}' + (':key16' if args.key16 else ''), {})
    except Exception:
        pmc = {}
    try:
        lib_sha = open(os.path.join(ROOT, 'mv2d_amd', 'lib', 'libmv2d_hip.so.sha256')).read().strip()
    except OSError:
        lib_sha = None
    pmc_stale = bool(pmc) and pmc.get('_csrc_sha256') != lib_sha       # the counters were collected on other kernel sources than the library that runs

    def roof(name, kernel, ms, flops, nbytes, launches=1):
        """roofline object of one kernel: achieved = algorithmic flops / bytes of ONE launch over its duration (HIP events on the launch
        stream); traffic = HBM bytes of one launch from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/pmc_traffic.json)."""
        tf, gb = flops / (ms * 1e-3) / 1e12, nbytes / (ms * 1e-3) / 1e9
        f_mfma, f_hbm = tf / PEAK_BF16_TFLOPS, gb / PEAK_HBM_GBS
        t = pmc.get(name)
        o = dict(bound='hbm' if f_hbm > f_mfma else 'mfma', kernel=kernel, launch_ms=round(ms, 4), launches_per_step=launches,
                 bytes_per_launch=int(nbytes), flops_per_launch=float(flops))
        if f_hbm > f_mfma:
            o.update(achieved=round(gb, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(f_hbm, 4), mfma_frac=round(f_mfma, 4))
        else:
            o.update(achieved=round(tf, 2), peak=PEAK_BF16_TFLOPS, unit='TFLOP/s', frac=round(f_mfma, 4), hbm_frac=round(f_hbm, 4))
        o['traffic'] = (t['fetch_bytes'] + t['write_bytes']) if t else None
        o['traffic_source'] = ('committed profile (profiles/pmc_traffic.json: rocprofv3 --pmc passes, tools/pmc_bench.sh); not measured in this run' +
                               ('; STALE: collected on other kernel sources than this library (csrc digest differs)' if pmc_stale else '; same csrc digest as this library')) if t else None
        o['traffic_stale'] = pmc_stale if t else None
        o['traffic_detail'] = dict(t, source='profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)') if t else None
        return o

    sk = dict(STAGE_KERNEL)
    if eng.exact:
        # index-exact route: the PE stage = pe_frustum_f32_kernel + pe_x3_kernel (3 MFMAs per product: `mfma_issue_frac` prices the issued MFMA work,
        # `frac` the algorithmic FLOPs as SURVEY 8(d) counts them)
        sk['pe_fused'] = 'pe_frustum_f32_kernel + pe_x3_kernel (frustum rows in fp64, frustum MLP + gate in split precision, sine branch from the table)'
        by['pe_fused'] = S * 192 * 4 * 2 + S * 256 * 4 * 2 + (192 * 1024 + 1024 * 256 + 2 * 256 * 256) * 4 + (S * 256 * (4 + 8) if kind == 'T' else S * 256 * 4)
    stage_roofline = {k: roof(k, f"{sk.get(k, 'gemm_bf16_kernel')}[{k}]", stage_ms[k], fl[k], by[k]) for k in fl if k in stage_ms}
    if eng.exact and 'pe_fused' in stage_roofline:
        stage_roofline['pe_fused']['mfma_issue_frac'] = round(3 * fl['pe_fused'] / (stage_ms['pe_fused'] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)

    # ---------------- decoder ms/iter (CrossAttentionBoxHead transformer on prepared inputs), hipGraph replay
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng._enqueue_decoder(ws, R)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
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
    # the same for ONE sample per call (the reference's call shape, DET/mv2d.py:143)
    decoder_ms_b1 = None
    if B > 1 and not args.brief and not args.no_parity_leg:
        o1 = eng.run(feat, props, metas)
        torch.cuda.synchronize()
        ws1, R1 = o1['ws'], ws_rows(o1)
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            eng._enqueue_decoder(ws1, R1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g1, capture_error_mode='thread_local'):
            eng._enqueue_decoder(ws1, R1)
        for _ in range(5):
            g1.replay()
        e0.record()
        for _ in range(50):
            g1.replay()
        e1.record()
        torch.cuda.synchronize()
        decoder_ms_b1 = e0.elapsed_time(e1) / 50

    # ---------------- the tile cross-attention kernel alone (HBM-bound gather): HIP events around 20 launches on the prepared buffers
    xattn = None
    if True:
        xk, xv = ws['xk_rows'], ws['xv_rows']
        q_ord = ws.get('q_order')                          # T path: blocks in the order of the queries' smallest key (as the engine launches it)
        xlo = dict(Xk_lo=ws['xk_lo'], Xv_lo=ws['xv_lo']) if (getattr(eng, 'exact', False) and ws.get('xk_lo') is not None) else {}      # index-exact route: hi + lo rows
        # the launch the engine makes per layer: the one-launch cross attention (S path: query map + tile attention + context map, csrc/xattn_fused.hip)
        # or the tile kernel of the three-launch form (T path)
        fused = (kind == 'S') if eng.fuse_xattn is None else bool(eng.fuse_xattn)
        if fused:
            W0 = eng.w
            x_launch = lambda: ops.xattn_fused(ws['q'], W0['ca_mapA0'], W0['ca_mapB0'], W0['ca_v_b0'], xk, xv, ws['row_ptr'], ws['col_idx'], out=ws['ctx'], R=R,      # noqa: E731
                                               empty_nan=eng.empty_nan, order=q_ord, **xlo)
        else:
            x_launch = lambda: ops.xattn_tile(ws['Qt'], xk, xv, ws['row_ptr'], ws['col_idx'], ws['zh'], R, empty_nan=eng.empty_nan, waves=eng.xattn_waves,      # noqa: E731
                                              order=q_ord, **xlo)
        for _ in range(3):
            x_launch()
        e0.record()
        for _ in range(20):
            x_launch()
        e1.record()
        torch.cuda.synchronize()
        x_ms_idle = e0.elapsed_time(e1) / 20
        # `frac` is priced at the LONGER of this event time (idle GPU) and the kernel's average duration in the committed rocprofv3 kernel trace of the
        # default bench command (four streams: what the timed loop really sees; profiles/r06_default_bench_<workload>_kernel_stats.txt, made by
        # tools/r06_evidence.sh with this library's kernels) -- round 5 quoted the idle figure, ~5 % kinder
        x_ms, x_ms_prof = x_ms_idle, None
        try:
            kname = 'xattn_fused_kernel' if fused else 'xattn_tile_kernel'
            for l_ in open(os.path.join(ROOT, 'profiles', f'r06_default_bench_{args.workload.replace("_", "")}_kernel_stats.txt')):
                if kname in l_:
                    x_ms_prof = float(l_.split()[-4]) * 1e-3          # columns: ... calls total_us avg_us min_us max_us pct
                    break
        except Exception:      # noqa: BLE001
            x_ms_prof = None
        if x_ms_prof is not None and not args.key16:
            x_ms = max(x_ms_idle, x_ms_prof)
        # algorithmic HBM bytes: every key row that some query reads, once (K and V, key16 = 2 B per element) + Qt in + z out.  Rows read by several queries
        # (T path: 2.9 per row) are counted once here — the repeats are L2 / Infinity Cache traffic; `gathered_bytes` counts them all.
        n_rows = min(nnz, S if kind == 'T' else R * 49)
        lo8 = bool(xlo) and xlo['Xk_lo'].dtype == torch.uint8
        b_el = (3 if lo8 else 4) if xlo else 2                                      # bytes per key / value element: fp16 hi + e4m3 lo (round 6), fp16 hi + fp16 lo, or one fp16
        row_b = 2 * 256 * b_el                                                      # K + V row
        own = 0 if fused else R * (16 * 256 * 2 + 8 * 256 * 4)                      # Qt in + z out: intermediates of the three-launch decomposition (none when fused)
        # SURVEY 8(d) per layer: read X_k, X_v (2 S C b) + query state in / out (2 Q C 4).  `frac` follows from these bytes ALONE (round 4 also
        # counted Qt / z, which exist only because the attention is split into three kernels: that figure stays as frac_incl_own_intermediates)
        x_bytes = n_rows * row_b + 2 * R * 256 * 4
        x_gathered = nnz * row_b + own
        x_flops = 2.0 * nnz * 8 * 256 * 2                                          # logits + P.V in the 256-dim input space, 8 heads
        xattn = roof('xattn_fused' if fused else 'xattn_tile',
                     ('xattn_fused_kernel (query map + sparse cross-attention in the raw key space + context map, ONE launch per decoder layer)' if fused else
                      'xattn_tile_kernel (sparse cross-attention in the raw key space, one launch per decoder layer; query / context maps are separate launches)'),
                     x_ms, x_flops, x_bytes, launches=eng.L)
        gbs = lambda nb: nb / (x_ms * 1e-3) / 1e9      # noqa: E731
        xattn['bytes_per_element'] = b_el
        xattn['launch_ms_idle_gpu'] = round(x_ms_idle, 4)
        xattn['launch_ms_rocprof_committed'] = None if x_ms_prof is None else round(x_ms_prof, 4)
        xattn['launch_ms_note'] = ('launch_ms (and frac) = the longer of launch_ms_idle_gpu (HIP events around 20 launches on their stream, idle GPU, this run) and '
                                   'launch_ms_rocprof_committed (average duration of the kernel in the committed rocprofv3 kernel trace of the default four-stream bench, profiles/)')
        xattn['frac_incl_own_intermediates'] = round(gbs(n_rows * row_b + own) / PEAK_HBM_GBS, 4)
        xattn['frac_at_survey_b2'] = round(gbs(n_rows * 2 * 256 * 2 + 2 * R * 256 * 4) / PEAK_HBM_GBS, 4)
        xattn['frac_at_b4'] = round(gbs(n_rows * 2 * 256 * 4 + 2 * R * 256 * 4) / PEAK_HBM_GBS, 4)
        xattn['gathered_bytes_per_launch'] = int(x_gathered)
        xattn['gathered_gbs'] = round(x_gathered / (x_ms * 1e-3) / 1e9, 1)
        xattn['note'] = ('bytes_per_launch = SURVEY 8(d): K and V rows read by at least one query, once (b bytes per element: 3 = fp16 hi + e4m3 lo of the '
                         'index-exact route since round 6 -- the bytes this kernel has to move; 4 = fp16 hi + lo pairs (lo8_rows = False, rounds 3-5), the width the '
                         'fp32 reference reads; 2 = key16 mode) + the query state in / out (2 Q C 4); frac_at_survey_b2 / frac_at_b4 = the same time priced at the '
                         'b = 2 SURVEY 8(d) assumed / at the fp32 reference\'s b = 4 (what rounds 3-5 quoted as frac); frac_incl_own_intermediates also counts Qt in + z out (8 KB each per query); '
                         'gathered_bytes_per_launch = the rows of every allowed (query, key) pair (repeats are served by L2 / Infinity Cache)')
        stage_roofline['xattn_fused' if fused else 'xattn_tile'] = xattn
    # the dominant kernel = the one with the most time per step (launch duration x launches per step)
    dom = max(stage_roofline, key=lambda k: stage_roofline[k]['launch_ms'] * stage_roofline[k]['launches_per_step'])
    roofline = stage_roofline[dom]

    # ---------------- CPU baseline: the oracle (port of the reference algorithm) on the host cores, bounded sample
    cpu, cpu2 = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import subprocess

        def cpu_leg(threads, iters, tmo, more=()):
            pr = subprocess.Popen([sys.executable, '-m', 'oracle.cpu_baseline', '--workload', args.workload, '--iters', str(iters),
                                   '--threads', str(threads), *more], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            try:
                so, se = pr.communicate(timeout=tmo)
                timed_out = False
            except subprocess.TimeoutExpired:
                pr.kill()
                so, se = pr.communicate()
                timed_out = True
            recs = []
            for l in (so or '').splitlines():
                if l.startswith('{'):
                    try:
                        recs.append(json.loads(l))
                    except ValueError:
                        pass
            final = [r_ for r_ in recs if 'value' in r_]
            if final:
                return final[-1]
            # the leg was cut off: report what its per-frame progress lines say (a frame time IS a measurement, however slow)
            fr = [r_['frame_s'] for r_ in recs if r_.get('progress') == 'timed'] or [r_['frame_s'] for r_ in recs if r_.get('progress') == 'warmup']
            if fr:
                med = statistics.median(fr)
                return dict(value=round(1.0 / med, 4), unit='samples/s', cores=threads, kind='port', frames_timed=len(fr),
                            sample=f'cut off after {tmo}s: median of the {len(fr)} frame(s) of {args.workload} that finished ({med:.1f} s per frame, {threads} threads)')
            return dict(value=None, unit='samples/s', cores=threads, kind='port',
                        sample=(f'not one frame of {args.workload} finished in {tmo}s with {threads} threads (< {1.0 / tmo:.4f} samples/s)' if timed_out
                                else f'oracle subprocess failed: {(se or "")[-200:]}'))
        # BASELINE.md section 3 asks for torch.set_num_threads(os.cpu_count()), 3 warm-ups + 10 timed frames.  On the 256-thread hosts of this pool the
        # oracle's many small operators spend their time waking threads, so the thread count that MAXIMISES the baseline is found first (a short sweep,
        # 3 warm-ups + 6 timed frames each, bounded) and the protocol's 3 + 10 frames (+ the decoder-only leg) run at that count; the all-cores leg is
        # bounded and reports what it measured, however slow.
        nc_ = os.cpu_count() or 1
        sweep = {}
        if args.cpu_threads > 0:
            sweep[args.cpu_threads] = None
        else:
            for th in (8, 16, 32, 64):
                if th <= nc_:
                    r_ = cpu_leg(th, 6, 60, ('--warmups', '3', '--budget-s', '20', '--no-decoder'))
                    sweep[th] = r_.get('value')
        best = max(sweep, key=lambda k: sweep[k] or 0.0) if sweep else min(nc_, 16)
        cpu = cpu_leg(best, args.cpu_iters, args.cpu_timeout, ('--warmups', '3'))
        if cpu is not None and len(sweep) > 1:
            cpu['thread_sweep_samples_s'] = {str(k): v for k, v in sweep.items()}
            cpu['threads_chosen'] = 'the count of the sweep with the highest samples/s'
        if nc_ != best:
            cpu2 = cpu_leg(nc_, 3, args.cpu_all_budget * 2 + 20, ('--warmups', '3', '--budget-s', str(args.cpu_all_budget), '--no-decoder'))

    other = None
    if rank == 0 and world == 1 and not args.no_other_workloads and args.workload == 'cfg2_s':
        # short legs of the T-path workloads in the same JSON line (sub-processes of this script, after this process' GPU work is done)
        import subprocess
        other = {}
        # cfg2_s_nc6: the headline size on the overlapping rig (mv2d_amd/synthetic.py RIG): the reference's own correlation gives every query its
        # RoI plus up to five matched ones (3.99 on average) instead of the 1.01 of the ring rig -- the non-trivial S workload
        for wl_, b_ in (('cfg2_s_nc6', 16), ('cfg3_t', 16), ('cfg5_t', 4)):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--workload', wl_, '--batch', str(b_), '--steps', '100', '--warmup', '10', '--rounds', '1',
                                    '--brief'], cwd=ROOT, capture_output=True, text=True, timeout=240)
                ls_ = [l for l in r.stdout.splitlines() if l.startswith('{')]
                d_ = json.loads(ls_[-1])
                other[wl_] = {k: d_.get(k) for k in ('value', 'unit', 'route', 'ms_per_step', 'steps', 'config', 'decoder_ms_per_iter', 'decoder_ms_per_launch', 'roofline',
                                                     'index_mismatches')}
                # ... and the same loop in the opt-in key16 mode
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--workload', wl_, '--batch', str(b_), '--steps', '60', '--warmup', '10', '--rounds', '1',
                                    '--brief', '--key16'], cwd=ROOT, capture_output=True, text=True, timeout=240)
                ls_ = [l for l in r.stdout.splitlines() if l.startswith('{')]
                d2_ = json.loads(ls_[-1])
                other[wl_]['samples_s_key16_mode_opt_in'] = d2_.get('value')
                other[wl_]['index_exact_vs_key16_mode'] = round(d_['value'] / d2_['value'], 3) if d2_.get('value') else None
                other[wl_]['index_mismatches_key16_mode'] = (d2_.get('index_mismatches') or {}).get('key16_mode')
            except Exception as ex_:          # noqa: BLE001
                other[wl_] = dict(value=None, error=repr(ex_)[:300])

    coll_leg = None
    if rank == 0 and world == 1 and not collective and not args.no_collective_leg:
        # RCCL on the one GPU there is: the same step with the process group initialised (nccl, one rank) and the per-step all-gather of the
        # decoded boxes in it (mv2d_amd.dist.gather_detections, what N > 1 ranks run), >= 200 steps under hipGraph replay, gathered == packed
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--workload', args.workload, '--batch', str(B), '--inflight', str(args.inflight),
                                '--rotate', str(args.rotate), '--rounds', str(G), '--steps', str(max(40, args.steps)), '--warmup', '5', '--brief', '--force-collective', '--no-parity-leg'],
                               cwd=ROOT, capture_output=True, text=True, timeout=300)
            ls_ = [l for l in r.stdout.splitlines() if l.startswith('{')]
            d_ = json.loads(ls_[-1])
            coll_leg = dict(samples_s=d_.get('value'), steps=d_.get('steps'), ms_per_step=d_.get('ms_per_step'), check=d_.get('collective_check'),
                            vs_value=round(d_['value'] / value, 4) if d_.get('value') else None)
        except Exception as ex_:          # noqa: BLE001
            coll_leg = dict(samples_s=None, error=repr(ex_)[:300])

    # integer parity of BOTH routes against the reference goldens for all workloads of this line, at the top level (ranked (query, class)
    # indices that differ / list length; the reference ranks sigmoid(cls).view(-1).topk(300), CB/coders/nms_free_coder.py:49-102)
    parity = None
    if rank == 0 and 'index_mismatches' in extra:
        def _cnt(d_):
            return None if not d_ else '%s/%s' % (d_.get('ranked_indices'), d_.get('of'))
        im_ = extra['index_mismatches']
        parity = {args.workload: dict(index_exact=_cnt(im_.get('index_exact')), key16_mode=_cnt(im_.get('key16_mode')))}
        for wl_, d_ in (other or {}).items():
            if isinstance(d_, dict) and d_.get('index_mismatches'):
                parity[wl_] = dict(index_exact=_cnt(d_['index_mismatches'].get('index_exact')), key16_mode=_cnt(d_.get('index_mismatches_key16_mode')))
        try:
            rn_ = np.load(os.path.join(ROOT, 'tests', 'golden', 'refnoise.npz'))
            parity['reference_vs_itself'] = {k[:-len('_pairwise_ranked_diff')]: '%d/300' % int(rn_[k].max()) for k in rn_.files if k.endswith('_pairwise_ranked_diff')}
            parity['reference_vs_itself_note'] = ('ranked indices that differ between runs of the UNMODIFIED reference on the same inputs under other intra-op thread '
                                                  'counts / oneDNN off (oracle/gen_golden_refnoise.py; <workload>_s<seed>): the resolution of "bit-exact"')
        except Exception:      # noqa: BLE001
            pass

    if rank == 0:
        line = {
            'metric': 'multi-view samples/sec (6-cam frames) through the MV2D RoI-head hot path',
            'value': round(value, 2), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'setup_steps': args.prime,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': (('f16 (ONE rounding of the key side: opt-in key16 mode) / f16x3 split precision (query side)' if args.key16 else
                       'f16x3: every operand an fp16 hi + lo pair, three MFMAs per product, fp32 accumulation (fp32-class, index-exact route; the lo halves of the key / value rows are STORED as e4m3 bytes)')), 'data': 'synthetic',
            'route': 'key16_mode_opt_in' if args.key16 else 'index_exact',
            'timed_seconds': round(elapsed, 3),
            'config': {'workload': f'{args.workload}: MV2D-{kind} head, {len(metas)} views {metas[0]["img_shape"][1]}x{metas[0]["img_shape"][0]}, '
                                   f'R={R} queries, S={S} key positions, nnz={nnz} allowed (q,k) pairs = {nnz / max(R, 1):.1f} keys per query' + (f' (totals of the {B} samples of a launch)' if B > 1 else '') + (f', corr_topk={args.corr_topk}' if args.corr_topk else '') + (f', forced n_c={args.force_nc}' if args.force_nc else ''),
                       'frames_per_step_per_gpu': args.inflight * G * B, 'global_batch': world * args.inflight * G * B,
                       'streams_per_gpu': args.inflight, 'launch_sequences_per_stream_and_step': G, 'samples_per_launch': B,
                       'feature_map_memory': 'contiguous NCHW' if args.nchw_input else 'channels_last (position-major rows, as mv2d_amd.plugin.neck writes them; logical NCHW)', 'pe_sine_branch': 'folded into a per-(weights, geometry) table, FLOPs not counted',
                       'parallelism': f'dp{world}', 'hipgraph': use_graph},
            'decoder_ms_per_iter': round(decoder_ms / B, 4), 'decoder_ms_per_launch': round(decoder_ms, 4),
            'decoder_ms_per_iter_batch1': round(decoder_ms_b1, 4) if decoder_ms_b1 is not None else (round(decoder_ms, 4) if B == 1 else None),
            'long_run': long_run, 'ranked_index_mismatches_vs_reference': parity, 'other_workloads': other,
            'collective_check': collective_check,
            'samples_s_with_collective': coll_leg.get('samples_s') if coll_leg else None, 'collective_leg': coll_leg,
            'stage_ms': {k: round(v, 4) for k, v in stage_ms.items()},
            'roofline': roofline, 'stage_roofline': stage_roofline,
            'cpu_baseline': cpu, 'cpu_baseline_all_cores': cpu2,
            **extra,
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
