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


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def _gpu_numa_nodes(sys_root='/sys'):
    """NUMA node of every AMD GPU in PCI-address order (the order HIP enumerates them in when no *_VISIBLE_DEVICES mask reorders),
    read from sysfs: nothing here touches the GPU runtime."""
    base = os.path.join(sys_root, 'bus', 'pci', 'devices')
    nodes = []
    try:
        for bdf in sorted(os.listdir(base)):
            d = os.path.join(base, bdf)
            try:
                if open(os.path.join(d, 'vendor')).read().strip() != '0x1002':
                    continue
                # 3D / display / processing accelerators (not 0x0300: a board's VGA function is no compute device)
                if not open(os.path.join(d, 'class')).read().strip().startswith(('0x0302', '0x0380', '0x1200')):
                    continue
                nodes.append(int(open(os.path.join(d, 'numa_node')).read().strip()))
            except (OSError, ValueError):
                continue
    except OSError:
        pass
    return nodes


def rank_cpus(local_rank, local_world, allowed=None, sys_root='/sys'):
    """The host cores of local rank `local_rank` of `local_world` on this node -> (sorted cpu list, numa node or None).
    If sysfs names a NUMA node for as many GPUs as there are local ranks, a rank gets the cores of ITS GPU's node (shared evenly
    with the other ranks of that node); otherwise the allowed cores are cut into `local_world` contiguous shares (sockets own
    contiguous core ranges, so neighbouring ranks stay on one socket).  Pure host logic (tests/test_host_cpu.py)."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    local_world = max(int(local_world), 1)
    local_rank = int(local_rank) % local_world
    # a *_VISIBLE_DEVICES mask may drop or reorder devices: sysfs order is then no longer HIP's order and LOCAL_RANK no device index --
    # no NUMA guess in that case, contiguous shares of the allowed cores instead
    masked = any(os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'))
    nodes = [] if masked else _gpu_numa_nodes(sys_root)
    if len(nodes) >= local_world and all(n >= 0 for n in nodes[:local_world]):
        node = nodes[local_rank]
        try:
            cpus = [c for c in _parse_cpulist(open(os.path.join(sys_root, 'devices', 'system', 'node', f'node{node}', 'cpulist')).read())
                    if c in set(allowed)]
        except OSError:
            cpus = []
        mates = [r for r in range(local_world) if nodes[r] == node]
        if len(cpus) >= len(mates):
            i, m = mates.index(local_rank), len(mates)
            lo, hi = i * len(cpus) // m, (i + 1) * len(cpus) // m
            return cpus[lo:hi], node
    if len(allowed) >= local_world:
        lo, hi = local_rank * len(allowed) // local_world, (local_rank + 1) * len(allowed) // local_world
        return allowed[lo:hi], None
    return allowed, None


def pin_rank(local_rank=None, local_world=None):
    """Pin this process (and every worker it forks later) to rank_cpus(...).  Call BEFORE anything touches the GPU and before
    pools are forked.  -> dict(cpus=count, first=.., last=.., numa_node=..) for the per-rank diagnostics; {} when a single rank
    owns the node (nothing to separate) or BUFFER_NO_PIN is set."""
    local_rank = int(os.environ.get('LOCAL_RANK', 0)) if local_rank is None else local_rank
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', 1))) if local_world is None else local_world
    if local_world <= 1 or os.environ.get('BUFFER_NO_PIN'):
        return {}
    cpus, node = rank_cpus(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return {}
    try:                                      # a process that already imported torch keeps the whole machine's thread count otherwise: the
        import sys                            # intra-op pool would oversubscribe this rank's share
        if 'torch' in sys.modules:
            sys.modules['torch'].set_num_threads(max(1, len(cpus)))
    except Exception:
        pass
    return dict(cpus=len(cpus), first=cpus[0], last=cpus[-1], numa_node=node)


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
