"""Multi-GPU glue: one process per GPU, samples sharded data-parallel.

The hot path exchanges nothing between ranks during a forward; the only collective of an evaluation step is the
collection of decoded boxes, which replaces the pickle/tmpdir collection of mmdet's ``multi_gpu_test``
(tools/test.py:249-250): ONE fixed-size ``all_gather_into_tensor`` per step (RCCL over xGMI on the GPU box,
gloo in the CPU tests) — payload [max_num * 11 + 1] fp32 per sample = (boxes 9 | score | label) rows + count.
"""
import os

import torch
import torch.distributed as dist

ROW = 11


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torch.distributed.run). Returns (rank, world, local_rank)."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # MV2D_FORCE_COLLECTIVE=1: initialise the process group even for one rank (exercises the RCCL path on a 1-GPU box)
    force = os.environ.get('MV2D_FORCE_COLLECTIVE', '0') == '1'
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_samples(num_samples, rank, world):
    """Sample indices of this rank: r, r + world, ... (DistributedSampler order of tools/test.py:189-217)."""
    return list(range(rank, num_samples, world))


def pack_detections(boxes, scores, labels, count, max_num=300):
    """(boxes [max_num,9], scores [max_num], labels [max_num] int64, count [1] int32) -> [max_num*11 + 1] fp32.
    (torch formulation: CPU / gloo tests; on the GPU the same payload comes from one launch of mv2d_pack_detections)"""
    rows = torch.cat([boxes[:max_num], scores[:max_num, None], labels[:max_num, None].to(boxes.dtype)], 1)
    valid = (torch.arange(max_num, device=boxes.device) < count.to(torch.int64)).to(boxes.dtype)[:, None]
    return torch.cat([(rows * valid).reshape(-1), count.to(boxes.dtype).reshape(1)])


def pack_detections_batch(boxes, scores, labels, count, max_num=300):
    """run_batch outputs (boxes [B,max_num,9], scores [B,max_num], labels [B,max_num] int64, count [B] int32) -> [B, max_num*11 + 1]."""
    rows = torch.cat([boxes[:, :max_num], scores[:, :max_num, None], labels[:, :max_num, None].to(boxes.dtype)], 2)
    valid = (torch.arange(max_num, device=boxes.device)[None, :] < count.to(torch.int64)[:, None]).to(boxes.dtype)[:, :, None]
    return torch.cat([(rows * valid).reshape(rows.shape[0], -1), count.to(boxes.dtype)[:, None]], 1)


def unpack_detections(payload, max_num=300):
    n = int(payload[-1].item())
    rows = payload[:-1].view(max_num, ROW)[:n]
    return rows[:, :9], rows[:, 9], rows[:, 10].to(torch.int64)


def gather_detections(payload):
    """payload [B, max_num*11+1] of this rank -> [world, B, max_num*11+1] on every rank (one collective)."""
    force = os.environ.get('MV2D_FORCE_COLLECTIVE', '0') == '1'
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return payload[None]
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(payload.shape), dtype=payload.dtype, device=payload.device)
    dist.all_gather_into_tensor(out.view(world * payload.shape[0], -1), payload.contiguous())
    return out
