"""Multi-rank path on CPU: world_size-2 gloo processes shard the samples with no data-path collective and exchange
the decoded boxes with ONE fixed-size all-gather per step (mv2d_amd/dist.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mv2d_amd import dist as mdist


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, w, _ = mdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    mine = mdist.shard_samples(5, rank, world)
    payloads = []
    for s in mine[:2]:
        n = 3 + s
        boxes = torch.zeros((300, 9)); boxes[:n] = float(s + 1)
        scores = torch.zeros(300); scores[:n] = 0.5
        labels = torch.zeros(300, dtype=torch.int64); labels[:n] = s
        payloads.append(mdist.pack_detections(boxes, scores, labels, torch.tensor([n], dtype=torch.int32)))
    out = mdist.gather_detections(torch.stack(payloads))
    res = []
    for rr in range(world):
        for b in range(out.shape[1]):
            bx, sc, lb = mdist.unpack_detections(out[rr, b])
            res.append((rr, b, bx.shape[0], float(bx.sum()), int(lb.sum())))
    q.put((rank, mine, res))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert got[0][1] == [0, 2, 4] and got[1][1] == [1, 3]
    assert got[0][2] == got[1][2]                                         # every rank sees the same gathered result
    by = {(rr, b): (n, s, l) for rr, b, n, s, l in got[0][2]}
    assert by[(0, 0)][0] == 3 and by[(0, 1)][0] == 5 and by[(1, 0)][0] == 4 and by[(1, 1)][0] == 6
    assert by[(1, 1)] == (6, 6 * 9 * 4.0, 6 * 3)


def test_single_process_gather_is_identity():
    p = torch.arange(2 * 3301, dtype=torch.float32).view(2, 3301)
    assert torch.equal(mdist.gather_detections(p)[0], p)


def test_batch_payload_equals_per_sample_payloads():
    """run_batch outputs ([B,max_num,...], count [B]) pack into the same rows as B single-sample payloads (the wire format the
    GPU path produces with one launch of mv2d_pack_detections)."""
    g = torch.Generator().manual_seed(5)
    B, M = 4, 300
    boxes = torch.randn(B, M, 9, generator=g); scores = torch.rand(B, M, generator=g)
    labels = torch.randint(0, 10, (B, M), generator=g); count = torch.tensor([300, 0, 1, 123], dtype=torch.int32)
    batch = mdist.pack_detections_batch(boxes, scores, labels, count)
    assert batch.shape == (B, M * 11 + 1)
    for b in range(B):
        assert torch.equal(batch[b], mdist.pack_detections(boxes[b], scores[b], labels[b], count[b:b + 1]))
        bx, sc, lb = mdist.unpack_detections(batch[b])
        n = int(count[b])
        assert bx.shape == (n, 9) and torch.equal(bx, boxes[b, :n]) and torch.equal(lb, labels[b, :n])


def test_csr_transpose_groups_pairs_by_key():
    """ops.csr_transpose (torch ops, device agnostic): the allowed (query, key) pairs grouped by key, stable inside a key."""
    from mv2d_amd import ops
    row_ptr = torch.tensor([0, 3, 3, 5, 9], dtype=torch.int32)
    col = torch.tensor([4, 1, 2, 1, 0, 2, 4, 1, 3], dtype=torch.int32)
    key_ptr, pair_idx, pair_row = ops.csr_transpose(row_ptr, col, 6)
    assert pair_row.tolist() == [0, 0, 0, 2, 2, 3, 3, 3, 3]
    assert key_ptr.tolist() == [0, 1, 4, 6, 7, 9, 9]
    assert pair_idx.tolist() == [4, 1, 3, 7, 2, 5, 8, 0, 6]


def _reduce_mean_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mv2d_amd import train
    q.put((rank, train._reduce_mean(10.0 if rank == 0 else 40.0)))
    dist.destroy_process_group()


def test_reduce_mean_of_positives_world2():
    """The loss's averaging factor is the mean number of positives over the ranks (mmdet reduce_mean, cross_attention_head.py:419-420)."""
    import torch.multiprocessing as mp
    from mv2d_amd import train
    assert train._reduce_mean(7) == 7.0                      # no process group: the local value
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_reduce_mean_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    assert got == {0: 25.0, 1: 25.0}


def _allreduce_grads_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mv2d_amd import train
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    x = torch.full((3, 8), float(rank + 1))
    net[:3](x).sum().backward()                     # the last layer has no gradient on any rank ...
    if rank == 1:
        net[3].weight.grad = torch.ones_like(net[3].weight)      # ... except its weight on rank 1
    local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    one = train.allreduce_gradients(net.parameters())
    got = [p.grad.clone() for p in net.parameters()]
    for p, g in zip(net.parameters(), local):
        p.grad = None if g is None else g.clone()
    many = train.allreduce_gradients(net.parameters(), bucket_bytes=64)
    same = all(torch.equal(a, p.grad) for a, p in zip(got, net.parameters()))
    q.put((rank, one, many, same, [g.tolist() for g in got], [None if g is None else g.tolist() for g in local]))
    dist.destroy_process_group()


def test_allreduce_gradients_world2():
    """One flat all-reduce per bucket; result = mean over the ranks, missing gradients count as zeros, bucket size does not change it."""
    import torch.multiprocessing as mp
    from mv2d_amd import train
    assert train.allreduce_gradients(torch.nn.Linear(2, 2).parameters()) == 0          # no process group: no-op
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_allreduce_grads_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in range(2))}
    for p in ps:
        p.join(timeout=60)
    assert res[0][1] == res[1][1] == 1 and res[0][2] == res[1][2] > 1 and res[0][3] and res[1][3]
    assert res[0][4] == res[1][4]                                                    # both ranks hold the same gradients
    for k, (a, b) in enumerate(zip(res[0][5], res[1][5])):
        za = torch.zeros_like(torch.tensor(res[0][4][k])) if a is None else torch.tensor(a)
        zb = torch.zeros_like(torch.tensor(res[0][4][k])) if b is None else torch.tensor(b)
        assert torch.allclose(torch.tensor(res[0][4][k]), (za + zb) / 2)


# ---- the N > 1 branch of bench.py's step (pack -> all-gather -> unpack, ping-pong payload buffers) on two gloo ranks ------------------------
class _FakeEngine:
    """Stands for HeadEngine in bench.build_step: deterministic decoded boxes per (rank, stream, call)."""

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


def _bench_step_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    mdist.init_from_env(backend='gloo')
    inflight, B = 2, 3
    engs = [_FakeEngine(rank, i) for i in range(inflight)]
    sets = [[(None, [None] * B, [None] * B)] for _ in range(inflight)]
    pay = [torch.zeros((inflight * B, 300 * 11 + 1)) for _ in range(2)]
    state = dict(gathered_ev=[None, None], step_no=0)
    step = bench.build_step(engs, [None] * inflight, sets, None, B, pay, state, collective=True, use_graph=False, cuda=False)
    res = []
    for it in range(3):
        out = step()                                  # [world, inflight * B, 3301]
        assert out.shape == (world, inflight * B, 300 * 11 + 1)
        rows = []
        for rr in range(world):
            for j in range(inflight * B):
                bx, sc, lb = mdist.unpack_detections(out[rr, j])
                rows.append((rr, j, bx.shape[0], round(float(bx[0, 0]), 3), int(lb[0])))
        res.append(rows)
    q.put((rank, res, state['step_no']))
    dist.destroy_process_group()


def test_bench_step_two_ranks_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2] == 3          # both ranks hold the same gathered detections; ping-pong advanced
    inflight, B = 2, 3
    for it, rows in enumerate(got[0][1]):
        for rr, j, n, v, lab in rows:
            i, b = divmod(j, B)
            assert n == (rr * 7 + i * 3 + it + b) % 300 + 1
            assert abs(v - (100.0 * rr + 10.0 * i + it + 0.1 * b)) < 1e-3 and lab == (it + b) % 10


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_spawn_gloo():
    """`python bench.py --gpus 2` WITHOUT a launcher re-executes itself under torch.distributed.run, one rank per GPU (round 4 asserted
    WORLD_SIZE == --gpus and died).  --spawn-check stops after the rendezvous (no GPU here): two gloo ranks find each other and rank 0
    prints the line."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--spawn-check'], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[-1])
    assert d == dict(spawn_check=True, world=2, ranks=[0, 1], backend='gloo')


def test_bench_two_ranks_end_to_end_gloo_stub_engine():
    """`python bench.py --gpus 2 --backend gloo --stub-engine --steps 2`: the self-spawn, the timed loop with its barriers and max over ranks, the
    per-step all-gather on ping-pong buffers and the JSON line, END TO END on two CPU ranks (round 6; the HIP engine is replaced by bench.StubEngine,
    the line is labelled and is not a measurement).  Checks: both ranks ran, every rank's slice of the gathered tensor is what it packed, all ranks hold
    the same gathered tensor, `value` is the weak-scaling aggregate world x frames per step x steps / max-over-ranks time."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--stub-engine', '--steps', '2', '--warmup', '1',
                        '--prime', '1', '--batch', '3', '--inflight', '2', '--rounds', '2'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[-1])
    assert d['stub_engine'] is True and d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak'
    assert d['config']['frames_per_step_per_gpu'] == 2 * 2 * 3 and d['config']['global_batch'] == 2 * 12
    c = d['collective_check']
    assert c['backend'] == 'gloo' and c['world'] == 2 and c['gathered_shape'] == [2, 12, 3301]
    assert c['gathered_equals_packed'] and c['every_rank_holds_the_same_gathered_tensor'] and c['payload_nonzero_entries'] > 0
    assert c['steps_with_collective'] == 4                                   # prime + warm-up + 2 timed steps
    assert abs(d['value'] - 24 * 2 / d['timed_seconds']) <= 1e-2 * d['value'] and abs(d['ms_per_step'] - d['timed_seconds'] / 2 * 1e3) < 1e-2
