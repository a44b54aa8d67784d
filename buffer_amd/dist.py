"""(e) Multi-GPU: fragment pairs are independent, so the evaluation shards pair indices over the ranks
(one process per GPU, interleaved i -> rank i mod W to balance scenes of unequal size) with no
data-path collective; the only exchange is ONE all_gather of the per-pair poses (+ pair ids) at the
end -- RCCL over xGMI on the GPU box (backend 'nccl'), gloo in the CPU tests."""
import os

import torch
import torch.distributed as dist


def init(local_rank):
    """One process per GPU under torchrun: -> (rank, world, device, comm_device).  Backend 'nccl' (= RCCL over xGMI) unless
    BUFFER_DIST_BACKEND=gloo, which runs the same sharding / gather logic with host tensors (several ranks may then share
    one GPU: the CPU-side tests of the N > 1 path)."""
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    backend = os.environ.get('BUFFER_DIST_BACKEND', 'nccl')
    local = local_rank % max(torch.cuda.device_count(), 1) if backend == 'gloo' else local_rank
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    return rank, world, dev, (dev if backend == 'nccl' else torch.device('cpu'))


def shard_indices(n_pairs, rank, world):
    """pair i is processed by rank i mod world."""
    return list(range(rank, n_pairs, world))


def gather_poses(local_ids, local_poses, n_pairs, device=None):
    """all ranks call; local_poses f32[k,4,4] for pair ids local_ids (k may differ by one between ranks).
    -> f32[n_pairs,4,4] on every rank, row i = pose of pair i (identity where nothing was reported)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    device = device or (local_poses.device if isinstance(local_poses, torch.Tensor) else 'cpu')
    cap = (n_pairs + world - 1) // world
    # one [cap, 17] block per rank: column 0 carries the pair id as the BITS of an int32 (exact for any id; a float would stop
    # at 2^24), columns 1..16 the pose; -1 marks an unused row
    buf = torch.zeros((cap, 17), dtype=torch.float32, device=device)
    idcol = torch.full((cap,), -1, dtype=torch.int32, device=device)
    k = len(local_ids)
    if k:
        idcol[:k] = torch.as_tensor(local_ids, dtype=torch.int32, device=device)
        buf[:k, 1:] = torch.as_tensor(local_poses, dtype=torch.float32, device=device).reshape(k, 16)
    buf[:, 0] = idcol.view(torch.float32)
    if world > 1:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        allbuf = torch.cat(parts)
    else:
        allbuf = buf
    out = torch.eye(4, dtype=torch.float32, device=device).repeat(n_pairs, 1, 1)
    ids = allbuf[:, 0].contiguous().view(torch.int32)
    valid = ids >= 0
    out[ids[valid].long()] = allbuf[valid, 1:].reshape(-1, 4, 4)
    return out


def broadcast_limits(limits, device='cpu'):
    """neighbourhood limits are calibrated once (rank 0) and shared, so every rank truncates identically."""
    t = torch.as_tensor(list(limits), dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=0)
    return [int(x) for x in t.cpu()]
