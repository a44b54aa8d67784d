"""(e) Multi-GPU: fragment pairs are independent, so the evaluation shards pair indices over the ranks
(one process per GPU, interleaved i -> rank i mod W to balance scenes of unequal size) with no
data-path collective; the only exchange is ONE all_gather of the per-pair poses at the
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


def gather_poses(local_ids, local_poses, n_pairs, device=None, extra=None, explicit_ids=False):
    """all ranks call; local_poses f32[k,4,4] for pair ids local_ids (k may differ by one between ranks; k = 0 is fine).
    -> f32[n_pairs,4,4] on every rank, row i = pose of pair i (identity where nothing was reported).
    extra: optional f32[k,E] per-pair payload riding in the same exchange (ground truth for rank 0's evaluator) ->
    (poses, extras f32[n_pairs,E], zeros where nothing was reported).
    ONE all_gather of a [cap, 16 + E] float block per rank.  Pair ids do not travel: rank r's rows are the pairs
    shard_indices(n_pairs, r, world) in order (checked locally); with explicit_ids=True (any id assignment) they travel in a
    separate int32 all_gather -- never as bits inside the float block."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    device = device or (local_poses.device if isinstance(local_poses, torch.Tensor) else 'cpu')
    cap = (n_pairs + world - 1) // world
    k = len(local_ids)
    if k > cap:
        raise ValueError(f'gather_poses: {k} local pairs but at most {cap} per rank for {n_pairs} pairs on {world} ranks')
    if not explicit_ids and list(local_ids) != shard_indices(n_pairs, rank, world):
        raise ValueError('gather_poses: local_ids are not shard_indices(n_pairs, rank, world); pass explicit_ids=True')
    E = 0 if extra is None else int(extra.shape[1])
    buf = torch.zeros((cap, 16 + E), dtype=torch.float32, device=device)
    if k:
        buf[:k, :16] = torch.as_tensor(local_poses, dtype=torch.float32, device=device).reshape(k, 16)
        if E:
            buf[:k, 16:] = torch.as_tensor(extra, dtype=torch.float32, device=device).reshape(k, E)
    if world > 1:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [buf]
    if explicit_ids:
        idcol = torch.full((cap,), -1, dtype=torch.int32, device=device)
        if k:
            idcol[:k] = torch.as_tensor(list(local_ids), dtype=torch.int32, device=device)
        if world > 1:
            idparts = [torch.empty_like(idcol) for _ in range(world)]
            dist.all_gather(idparts, idcol)
        else:
            idparts = [idcol]
        id_lists = [[int(v) for v in ip.cpu().tolist() if v >= 0] for ip in idparts]
    else:
        id_lists = [shard_indices(n_pairs, r, world) for r in range(world)]
    out = torch.eye(4, dtype=torch.float32, device=device).repeat(n_pairs, 1, 1)
    ext = torch.zeros((n_pairs, E), dtype=torch.float32, device=device) if E else None
    for r, ids in enumerate(id_lists):
        if ids:
            idx = torch.as_tensor(ids, dtype=torch.long, device=device)
            out[idx] = parts[r][:len(ids), :16].reshape(-1, 4, 4)
            if E:
                ext[idx] = parts[r][:len(ids), 16:]
    return (out, ext) if E else out


def broadcast_limits(limits, device='cpu'):
    """neighbourhood limits are calibrated once (rank 0) and shared, so every rank truncates identically."""
    t = torch.as_tensor(list(limits), dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=0)
    return [int(x) for x in t.cpu()]
