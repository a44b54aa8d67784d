#!/usr/bin/env python3
"""Registration throughput of the MI355X-native BUFFER inference path (BASELINE.json metric:
registration pairs/sec; workload = configs[1], one 3DMatch-shape fragment pair, full inference, fp32).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step registers `--pairs-per-step` device-resident synthetic pairs on every rank through ONE set of
stacked launches per stage (`--mode batch`; `--mode threads` runs them on separate streams instead).
Pairs are sharded over the ranks with no data-path collective; one all_gather of the poses ends the
timed region.  Rank 0 prints
one JSON line (contract in the task description): whole-job pairs/s, the roofline of the dominant
hand-written kernel (k_grid_query, timed with HIP events on its launch stream inside the library)
and, at N=1, the CPU baseline (reference cpp_wrappers cores when oracle/_ref is built, else the
plain-C port, + the torch-CPU restatement of the model stages) timed on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from dataclasses import replace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COST_NET_FLOPS_PER_MATCH = 159994880.0   # csrc/costnet.hip (0.160 GFLOP/match)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_16x16x4_f32), 64 FLOP/clk/SIMD


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--keypts', type=int, default=5000, help='keypoints per fragment (BASELINE: ~5k)')
    ap.add_argument('--pairs-per-step', type=int, default=32, help='pairs registered per GPU and step')
    ap.add_argument('--mode', choices=['batch', 'threads'], default='batch',
                    help='batch: the pairs of a step share one set of stacked launches; threads: one stream per pair')
    ap.add_argument('--streams', type=int, default=1,
                    help='batch mode: the pairs of a step are split into this many stacked batches, one host thread + HIP '
                         'stream each, so that the latency-bound FPS of one batch overlaps the CNN kernels of another')
    ap.add_argument('--distinct-pairs', type=int, default=4, help='synthetic pairs generated per rank (cycled)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-keypts', type=int, default=256, help='keypoint sample of the CPU baseline leg')
    return ap.parse_args()


def cpu_baseline(sample, cfg, limits, keypts_full, keypts_sample):
    """One pair through the CPU path on this box's host cores.  The descriptor/matching stages run on
    a `keypts_sample`-keypoint sample and are scaled linearly to `keypts_full` (they are linear in the
    number of patches / matches); pyramid, point learner and FPS run in full."""
    from oracle import cpu, pipeline_ref
    from buffer_amd.weights import load_weights
    cpu.build(ref=True)
    use_ref = cpu.have_ref()
    W = {k: torch.from_numpy(v) for k, v in load_weights(cfg.weights).items()}
    rng = np.random.default_rng(0)
    perms = [rng.permutation(len(sample['src_fds_pts'])), rng.permutation(len(sample['tgt_fds_pts']))]
    tm = {}
    t0 = time.perf_counter()
    pipeline_ref.register_pair(sample, W, limits, cfg, 0, perms, num_keypts=keypts_sample, use_ref=use_ref, timings=tm)
    wall = time.perf_counter() - t0
    scale = keypts_full / keypts_sample
    # FPS is run for keypts_sample rounds only: scale it too (rounds are identical work)
    est = tm['pyramid'] + tm['point_learner'] + (tm['keypoints'] + tm['descriptors'] + tm['matching']) * scale + tm['pose']
    return dict(value=1.0 / est, unit='pairs/s', cores=torch.get_num_threads(),
                kind='reference' if use_ref else 'port',
                sample=f'1 pair: pyramid (cpp_wrappers cores, 1 thread) + point learner in full; FPS/descriptor/matching '
                       f'stages on {keypts_sample} of {keypts_full} keypoints per fragment, scaled x{scale:.1f}; '
                       f'torch-CPU {torch.get_num_threads()} threads; measured {wall:.1f} s',
                stages_s={k: round(v, 3) for k, v in tm.items()})


def main():
    a = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the documented torch.distributed.run command as a CHILD
        # (nothing has touched the GPU yet in this process; never re-exec) and hand its exit code on.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with\n  python -m torch.distributed.run --nnodes=1 '
                         f'--nproc-per-node {a.gpus} --master-addr 127.0.0.1 --master-port <P> bench.py --gpus {a.gpus} ...')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the product has no CPU path)')
    local = local % torch.cuda.device_count()      # (only matters for the single-GPU gloo smoke run below)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    backend = os.environ.get('BENCH_BACKEND', 'nccl')   # 'gloo' = code-path smoke test of the N>1 logic on one GPU
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == 'nccl' else torch.device('cpu')

    from buffer_amd import _lib, synth
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.pipeline import BufferPipeline
    cfg = replace(THREEDMATCH, num_keypts=a.keypts)
    pipe = BufferPipeline(cfg, dev)
    calib = synth.make_pair(1000)                     # same calibration pair on every rank -> identical limits
    limits = pipe.calibrate([calib])
    samples = [synth.make_pair(2000 + rank * 97 + i) for i in range(a.distinct_pairs)]
    inputs = [pipe.upload(s) for s in samples]
    torch.cuda.synchronize()
    L = _lib.lib()

    # pairs of one step run concurrently, one host thread + one HIP stream each: the latency-bound stages of
    # one pair (FPS occupies 2 CUs for milliseconds) overlap the chip-filling stages of another
    import threading
    from concurrent.futures import ThreadPoolExecutor
    nconc = max(1, a.pairs_per_step) if a.mode == 'threads' else max(1, min(a.streams, a.pairs_per_step))
    streams = [torch.cuda.Stream(device=dev) for _ in range(nconc)]
    tls = threading.local()
    slot_lock = threading.Lock()
    free_slots = list(range(nconc))

    def _bind():
        if not hasattr(tls, 'slot'):
            with slot_lock:
                tls.slot = free_slots.pop()
            torch.cuda.set_device(local)
        return streams[tls.slot]

    def one_pair(k):
        with torch.cuda.stream(_bind()):
            return pipe.register(inputs[k], seed=k)

    def one_batch(ks):
        with torch.cuda.stream(_bind()):
            return pipe.register_batch([inputs[k] for k in ks], seeds=ks)

    pool = ThreadPoolExecutor(max_workers=nconc) if nconc > 1 else None

    def step(i):
        ks = [(i * a.pairs_per_step + j) % len(inputs) for j in range(a.pairs_per_step)]
        if a.mode == 'batch':
            if pool is None:
                return pipe.register_batch([inputs[k] for k in ks], seeds=ks)
            parts = [ks[j::nconc] for j in range(nconc)]
            outs = list(pool.map(one_batch, parts))
            poses = [None] * len(ks)
            for j, o in enumerate(outs):
                poses[j::nconc] = o
            return poses
        if pool is None:
            return [pipe.register(inputs[k], seed=k) for k in ks]
        return list(pool.map(one_pair, ks))

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    L.buf_timing_enable(1)
    t0 = time.perf_counter()
    all_poses = []
    for i in range(a.steps):
        all_poses += step(i)
    mine = torch.stack(all_poses).to(cdev)
    if dist:                                           # the path's one exchange: poses of every shard
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L.buf_timing_enable(0)
    timed = {}
    for kid, name in ((0, 'grid_query'), (1, 'cyl_net'), (2, 'cost_net')):
        ms, work = C.c_double(0), C.c_double(0)
        n = L.buf_timing_collect_kernel(kid, C.byref(ms), C.byref(work))
        timed[name] = (int(n), ms.value, work.value)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # registration quality on this rank's pairs (DGR criterion of ThreeDMatch/test.py:264-270)
    ok = 0
    for n, pose in enumerate(all_poses):
        gt = samples[n % len(samples)]['relt_pose']
        T = pose.cpu().numpy().astype(np.float64)
        rte = np.linalg.norm(T[:3, 3] - gt[:3, 3])
        rre = np.degrees(np.arccos(np.clip((np.trace(T[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1)))
        ok += int(rte < 0.3 and rre < 15)

    if rank == 0:
        pairs = world * a.steps * a.pairs_per_step
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        pmc = json.load(open(tpath)) if os.path.exists(tpath) else {}
        per_launch = a.pairs_per_step / nconc if a.mode == 'batch' else 1      # pairs covered by one launch

        def roof(name, label, bound, peak, unit, scale, traffic):
            n, ms, work = timed[name]
            ach = (work / n) / (ms / n * 1e-3) / scale if n else 0.0
            return {'kernel': label, 'bound': bound, 'achieved': ach, 'peak': peak, 'unit': unit, 'frac': ach / peak,
                    'traffic': traffic, 'launches': n, 'avg_us': (ms / n * 1e3) if n else None,
                    'avg_algorithmic_' + ('bytes' if bound == 'hbm' else 'flops'): (work / n) if n else None}

        # HBM bytes per launch measured offline with rocprofv3 --pmc (profiles/traffic.json, separate passes)
        t_cyl = pmc.get('k_cyl_net_hbm_bytes_per_patch')
        t_cost = pmc.get('k_cost_net_hbm_bytes_per_match')
        t_grid = pmc.get('k_grid_query_hbm_bytes_per_launch_per_pair')
        npatch = 2 * a.keypts * per_launch
        gpu_ms = {k: v[1] / a.steps for k, v in timed.items()}
        out = {
            'metric': 'registration pairs/sec', 'value': pairs / elapsed, 'unit': 'pairs/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'one 3DMatch-shape fragment pair, full BUFFER inference (BASELINE configs[1])',
                       'pairs_per_step_per_gpu': a.pairs_per_step, 'step_mode': a.mode, 'streams': nconc, 'keypoints_per_fragment': a.keypts,
                       'fds_points': [int(s['src_fds_pts'].shape[0]) for s in samples[:1]]
                       + [int(s['tgt_fds_pts'].shape[0]) for s in samples[:1]],
                       'sds_points': [int(x) for x in inputs[0]['lengths']], 'neighbor_limits': limits,
                       'weights': '3DMatch 06132318 (released)', 'parallelism': f'pair-sharded x{world}',
                       'registered_ok': f'{ok}/{len(all_poses)} (rank 0, RTE<0.3 m & RRE<15 deg)'},
            # the dominant kernel of the step (k_cyl_net: ~60 % of the GPU time, profiles/r01_kernel_stats.csv)
            'roofline': roof('cyl_net', 'k_cyl_net (A11 Cylindrical_Net, fused fp32 MFMA)', 'mfma', MFMA_F32_PEAK_TFLOPS, 'TFLOP/s',
                             1e12, t_cyl * npatch if t_cyl else None),
            'roofline_other': [
                roof('cost_net', 'k_cost_net (A13 CostVolume + CostNet, fused fp32 MFMA)', 'mfma', MFMA_F32_PEAK_TFLOPS, 'TFLOP/s',
                     1e12, t_cost * timed['cost_net'][2] / max(timed['cost_net'][0], 1) / COST_NET_FLOPS_PER_MATCH if t_cost else None),
                roof('grid_query', 'k_grid_query (A2 radius neighbours)', 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9,
                     t_grid * per_launch if t_grid else None)],
            'timed_kernel_ms_per_step': gpu_ms,
        }
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(samples[0], cfg, limits, a.keypts, a.cpu_keypts)
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
