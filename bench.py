#!/usr/bin/env python3
"""Registration throughput of the MI355X-native BUFFER inference path (BASELINE.json metric:
registration pairs/sec; default workload = configs[1], one 3DMatch-shape fragment pair, full inference, fp32,
~5k keypoints per fragment).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A step registers `--pairs-per-step` device-resident synthetic pairs on every rank through ONE set of stacked launches
per stage.  Pairs are sharded over the ranks with no data-path collective; one all_gather of the poses ends the timed
region (barrier + synchronize on both sides, max over ranks).  Rank 0 prints ONE JSON line: whole-job pairs/s, the
`roofline` object of the dominant hand-written kernel (k_cyl_net_wg; HIP events on its launch stream inside the library),
`roofline_other` for every other kernel SURVEY 8(d) assigns a roofline class to, and, at N=1, `cpu_baseline`: the
reference's cpp_wrappers cores (oracle/_ref) + the torch-CPU restatement of the model timed on this box's host cores on
the SAME pair at the SAME keypoint count (no extrapolation).

Other workloads (each prints its own line with the same metric; they are not the headline):
  --workload stream   BASELINE configs[2]: 1623 synthetic pairs from RAW clouds, pre-processing included, with DGR recall + RR
  --workload kitti    BASELINE configs[3]: KITTI-shape ring scans (~120k returns, 0.05 / 0.30 m voxels), KITTI constants
  --workload surface  the DROP-IN mode: one 3DMatch-shape pair at 1500 keypoints, every operator of the reference's import surface
                      called THROUGH THE SHIM PACKAGES in the reference's own call order, shapes and host conventions (numpy in / out
                      for cpp_wrappers, batch 1, un-fused ball_query + grouping_operation), beside the same call on the host cores
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
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_16x16x4_f32) = fp32 vector peak
COST_NET_DENSE_FLOPS_PER_MATCH = 159994880.0      # 0.160 GFLOP/match, every layer as a dense convolution
COST_NET_FLOPS_PER_MATCH = 51905536.0             # executed by csrc/costnet.hip: layer 0 separated into its S- and T-terms,
#                                                   layers 1..5 in the Winograd domain (16 products per 2 x 2 output tile instead of 36)
CYL_NET_EXECUTED_FRACTION = 40 * 1024 / (9 * 140 * 4 * 16)   # k_cyl_net_wg runs the stack in the Winograd F(2x2,3x3) domain:
#                                          40 MFMAs of 16x16x4 per (4 input channels, 16 output channels) instead of the
#                                          9 x 140 x 4 x 16 MACs of the direct form = 0.508 of the dense count

MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA (v_mfma_f32_16x16x32_f16), ~2.5 PFLOP/s
VALU_ISSUE_PEAK_GINSTR = 256 * 4 * 2.4 / 4          # G wave-instructions/s: 1024 SIMDs, one 64-lane VALU instruction per 4 cycles, 2.4 GHz
# SQ_INSTS_VALU per patch of k_patch_voxelize and the library version it was counted on: profiles/instr.json (tools/make_instr.py)
CYL_NET_DENSE_FLOPS_PER_PATCH = 118702080.0       # 2 x 140 x sum 9 Cin Cout
CYL_NET_WIDTHS = (48, 64, 64, 128, 128, 64, 64, 32, 32)       # Cylindrical_Net channels (patchnet.py:15-85; layer 0 = the 3 x 16 radial x point-MLP planes)


def cyl_net_split_mfmas_per_patch(widths=CYL_NET_WIDTHS):
    """csrc/convnet_h3.hip issues 3 v_mfma_f32_16x16x32_f16 per (tap, 16 outputs, 16 positions, 32 channels): 9 taps, 9 position
    tiles for the 140 positions, k-steps of 32 input channels (layer 0's 48 as two): 22842 matrix instructions per patch."""
    return sum(3 * 9 * 9 * (co // 16) * ((ci + 31) // 32) for ci, co in zip(widths[:-1], widths[1:]))


CYL_NET_SPLIT_ISSUED_FLOPS_PER_PATCH = cyl_net_split_mfmas_per_patch() * 16384.0


def load_instr():
    p = os.path.join(ROOT, 'profiles', 'instr.json')
    return json.load(open(p)) if os.path.exists(p) else {}

# library timing ids (include/buffer_hip.h BUF_TIMED_*)
TIMED = {'grid_query': 0, 'cyl_net': 1, 'cost_net': 2, 'select_patches': 3, 'patch_voxelize': 4, 'fps': 5, 'nn1': 6,
         'vn_gather': 7, 'grid_subsample': 8, 'desc_head': 9, 'cyl_net_split': 10, 'cost_net_split': 11}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['pair', 'stream', 'kitti', 'surface'], default='pair')
    ap.add_argument('--keypts', type=int, default=None,
                    help='keypoints per fragment (pair: 5000 = BASELINE; stream / kitti: 1500 = the reference configs)')
    ap.add_argument('--pairs-per-step', type=int, default=None, help='pairs registered per GPU and step (pair: 32, kitti: 16)')
    ap.add_argument('--streams', type=int, default=1,
                    help='the pairs of a step are split into this many stacked batches, one host thread + HIP stream each')
    ap.add_argument('--no-pipeline', action='store_true',
                    help='do not overlap the keypoint stage of step i+1 with the descriptor stage of step i (two HIP streams)')
    ap.add_argument('--distinct-pairs', type=int, default=None,
                    help='synthetic pairs generated per rank, distinct seeds (default: pairs-per-step, so a step never repeats a pair)')
    ap.add_argument('--stream-pairs', type=int, default=1623, help='1623 = 3DMatch test set, 1781 = 3DLoMatch')
    ap.add_argument('--stream-overlaps', default=None,
                    help='overlap classes of the synthetic stream, equal shares (default 0.75,0.6,0.45,0.3; a 3DLoMatch-like set: 0.3,0.25,0.2,0.15)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--arith', choices=['f32', 'split'], default='f32',
                    help="CNN kernels of the (first) timed region: 'f32' = fp32 MFMA (default, the headline), 'split' = fp32-equivalent split-f16")
    ap.add_argument('--no-split', action='store_true',
                    help="skip the second timed region with cnn_arith='split' (fp32-equivalent split-f16 CNN kernels: value_split, roofline_split)")
    ap.add_argument('--total-pairs', type=int, default=None,
                    help='pair workload, STRONG scaling: this many pairs per step for the whole job, pair i of a step on rank i mod N '
                         '(default: --pairs-per-step on every rank = weak scaling)')
    ap.add_argument('--detail-json', default=None, help='also write the uncompacted record (full roofline_other entries) to this file')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(sample, cfg, limits, hip_register=None, hip_register_split=None):
    """The CPU path on THIS box's host cores, one pair, full size (no sampling of keypoints, nothing extrapolated):
    pyramid = the reference's cpp_wrappers cores (oracle/_ref, kind 'reference'; the plain-C port where that library was
    not built) on min(16, nproc) concurrent workers like the reference's DataLoader (ThreeDMatch/config.py:22), each
    building the pyramid of one pair; model stages = torch-CPU restatement on all threads.  Baseline only.
    hip_register / hip_register_split(perms, seed) -> (pose, detail): the product path (fp32 / split-f16 CNN kernels) on the SAME pair
    with the SAME pinned permutations and seed -- the oracle's result is not thrown away but compared with them: `parity`,
    `parity_split` (the checker role of oracle/, at the full BASELINE size)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cpu, pipeline_ref, torch_ref
    from buffer_amd.weights import load_weights
    cpu.build(ref=True)
    use_ref = cpu.have_ref()
    W = {k: torch.from_numpy(v) for k, v in load_weights(cfg.weights).items()}
    rng = np.random.default_rng(0)
    perms = [rng.permutation(len(sample['src_fds_pts'])), rng.permutation(len(sample['tgt_fds_pts']))]
    workers = min(16, os.cpu_count() or 1)

    def pyramid(_):
        return torch_ref.collate(sample, limits, cfg.voxel_size_0, cfg.conv_radius, use_ref)

    pyramid(0)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(workers) as ex:        # ctypes releases the GIL: the KD-tree / grid cores run truly concurrently
        list(ex.map(pyramid, range(workers)))
    t_pyr = (time.perf_counter() - t0) / workers   # seconds per pair at `workers` loaders
    tm = {}
    t0 = time.perf_counter()
    want, wd = pipeline_ref.register_pair(sample, W, limits, cfg, 0, perms, use_ref=use_ref, timings=tm)
    t_model = sum(v for k, v in tm.items() if k != 'pyramid')

    def compare(fn):
        got, gd = fn(perms, 0)
        kp_equal = all(np.array_equal(gd['kpts'][i].cpu().numpy(), wd['kpts'][i].numpy()) for i in range(2))
        # descriptors row by row: a point of a patch that sits ON a voxel ball's surface can fall on either side when the aligned
        # patch coordinates differ in the last bit (torch's CPU matmul vs the kernel's Rodrigues product): a handful of rows in 10^4
        # see another sample there (tools/desc_diff_probe.py); every other row agrees to fp32 round-off
        ddesc = torch.cat([(gd['desc'][i]['desc'].cpu() - wd['desc'][i]['desc']).abs().amax(1) for i in range(2)]) if kp_equal else None
        # ... and the cause is SHOWN for every such row (round 6, buffer_amd/diagnose.py): the two aligned patches agree to the accuracy of
        # two fp32 Rodrigues rotations (4e-6), the fp32 hit masks of the 420 x 512 (centre, point) pairs differ in at least one pair, and
        # every differing pair's margin |d^2 - r^2| is within what the observed coordinate difference of that point can move.  A row over tolerance WITHOUT that is counted as
        # unexplained (the bench-contract test requires 0).
        over, unexplained, shown = [], 0, []
        if kp_equal:
            from buffer_amd import diagnose
            from buffer_amd.patch_embedder import voxel_centres
            centres = voxel_centres(cfg.rad_n, cfg.azi_n, cfg.ele_n)
            P = cfg.num_keypts
            for row in torch.nonzero(ddesc > 1e-4).flatten().tolist():
                c, k = divmod(row, P)
                e = diagnose.explain_row(gd['desc'][c]['patches'][k].cpu().numpy(), centres, cfg.delta / cfg.rad_n,
                                         theirs=wd['desc'][c]['patches'][k].numpy())
                over.append(row)
                unexplained += 0 if e['explained'] else 1
                if len(shown) < 3:
                    shown.append(dict(cloud=c, keypoint=k, desc_abs_diff=float(ddesc[row]), **e))
        mine = set(zip(gd['s_mids'].cpu().numpy().tolist(), gd['t_mids'].cpu().numpy().tolist()))
        ref = set(zip(np.asarray(wd['s_mids']).tolist(), np.asarray(wd['t_mids']).tolist()))
        return dict(keypoints_equal=bool(kp_equal), matches=len(ref), matches_differing=len(mine ^ ref),
                      pose_max_abs_diff=float(np.abs(got.cpu().numpy().astype(np.float64) - want.astype(np.float64)).max()),
                      desc_rows=None if ddesc is None else int(ddesc.numel()),
                      desc_rows_differing_over_1e_4=None if ddesc is None else int((ddesc > 1e-4).sum()),
                      desc_rows_over_1e_4_not_explained_by_a_ball_surface_flip=None if ddesc is None else int(unexplained),
                      desc_rows_over_1e_4_examples=shown,
                      desc_max_abs_diff_other_rows=None if ddesc is None else float(ddesc[ddesc <= 1e-4].max()),
                      desc_max_abs_diff=None if ddesc is None else float(ddesc.max()),
                      what=f'HIP register() vs oracle/pipeline_ref.register_pair on this pair at {cfg.num_keypts} keypoints/fragment, '
                           'same permutations, seed 0')

    parity = compare(hip_register) if hip_register is not None else None
    parity_split = compare(hip_register_split) if hip_register_split is not None else None
    if parity_split is not None:
        parity_split['what'] = "the same comparison for cnn_arith='split' (the fp32-equivalent split-f16 CNN kernels)"
    # the reference overlaps its loader workers with the model process: steady-state rate = the slower of the two legs
    return dict(value=1.0 / max(t_pyr, t_model), unit='pairs/s', cores=torch.get_num_threads(), workers=workers,
                kind='reference cores (pyramid) + restated model' if use_ref else 'port',
                sample=f'1 pair at full size ({cfg.num_keypts} keypoints/fragment, nothing scaled): pyramid by the '
                       f'{"reference cpp_wrappers cores" if use_ref else "plain-C port"} on {workers} workers ({t_pyr * 1e3:.1f} ms/pair), '
                       f'model stages = torch-CPU restatement, {torch.get_num_threads()} threads ({t_model:.1f} s/pair); '
                       f'value = 1 / max(loader leg, model leg)',
                stages_s={k: round(v, 3) for k, v in tm.items()}, pyramid_s_per_pair_at_workers=round(t_pyr, 4), parity=parity,
                **({'parity_split': parity_split} if parity_split is not None else {}))


# ------------------------------------------------------------------------------------------------ helpers
def collect_timed(L):
    out = {}
    for name, kid in TIMED.items():
        ms, work = C.c_double(0), C.c_double(0)
        n = L.buf_timing_collect_kernel(kid, C.byref(ms), C.byref(work))
        out[name] = (int(n), ms.value, work.value)
    return out


def load_traffic():
    if os.environ.get('BUF_NO_TRAFFIC'):            # the profiled runs of tools/profile_round.sh: no stale bytes in their lines
        return {}
    p = os.path.join(ROOT, 'profiles', 'traffic.json')
    return json.load(open(p)) if os.path.exists(p) else {}


def roof_entry(timed, name, label, bound, peak, unit, scale, traffic=None, **extra):
    n, ms, work = timed[name]
    if not n:
        return None
    ach = (work / n) / (ms / n * 1e-3) / scale
    e = {'kernel': label, 'bound': bound, 'achieved': ach, 'peak': peak, 'unit': unit, 'frac': ach / peak, 'traffic': traffic,
         'launches': n, 'avg_us': ms / n * 1e3, 'avg_algorithmic_' + ('bytes' if unit == 'GB/s' else 'flops'): work / n}
    e.update(extra)
    return e


def voxelize_entry(timed, traffic, patches, lib_version=None):
    """k_patch_voxelize is vector-ALU issue bound (SQ counters, profiles/r04_voxelize_sq.txt: the vector ALUs are busy 94 % of the
    kernel's cycles): achieved = wave-level VALU instructions per second (the measured count per patch x this launch's patches /
    HIP-event time), peak = 1024 SIMDs x one wave instruction per 4 cycles at 2.4 GHz.  The byte rate rides along."""
    e = roof_entry(timed, 'patch_voxelize', 'k_patch_voxelize (A9 + A10 + point MLP)', 'valu', HBM_PEAK_GBS, 'GB/s', 1e9, traffic)
    instr = load_instr()
    per_patch = instr.get('kernels', {}).get('k_patch_voxelize', {}).get('valu_per_patch')
    if not e or not patches or not per_patch:
        return e
    e['hbm_gbps'], e['hbm_frac'] = e['achieved'], e['frac']
    e['valu_wave_instructions_per_patch'] = per_patch
    e['achieved'] = per_patch * patches / (e['avg_us'] * 1e-6) / 1e9
    e['peak'], e['unit'] = VALU_ISSUE_PEAK_GINSTR, 'G wave-instructions/s'
    e['frac'] = e['achieved'] / e['peak']
    e['source'] = 'SQ_INSTS_VALU / patches, profiles/instr.json (tools/make_instr.py)'
    e['instr_lib_version'] = instr.get('lib_version')
    # current = counted on THIS library version AND on this very voxelize.hip (kernels changed without a version bump)
    import hashlib
    src = os.path.join(ROOT, 'buffer_amd', 'csrc', 'voxelize.hip')
    sha = hashlib.sha256(open(src, 'rb').read()).hexdigest()[:16] if os.path.exists(src) else None
    same_src = instr.get('csrc_sha256_16', {}).get('voxelize.hip') == sha
    e['instr_current'] = None if lib_version is None else bool(instr.get('lib_version') == lib_version and same_src)
    return e


def nn1_entry(timed):
    """A12 since round 4: k_nn1f_sweep ranks all pairs with 3 v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block in each of its two
    passes (min, then candidates) and forms the reference's fp32 sum for the candidates only: achieved = ISSUED f16 flops
    (6 x the 2 b q n 32 of the distance form) / HIP-event time of the whole call (prep + sweep + fallback launch)."""
    e = roof_entry(timed, 'nn1', 'k_nn1f_sweep (A12 mutual 1-NN: split-f16 MFMA ranking of all pairs, fp32 sum of the candidates)', 'mfma',
                   MFMA_F16_PEAK_TFLOPS, 'TFLOP/s', 1e12)
    if e:
        e['achieved'] *= 6.0
        e['frac'] = e['achieved'] / e['peak']
        e['avg_issued_flops'] = 6.0 * e['avg_algorithmic_flops']
    return e


def traffic_of(pmc, kernel, units):
    """HBM bytes per launch from the committed PMC summary (profiles/traffic.json: bytes per unit of work, measured with
    separate --pmc FETCH_SIZE / WRITE_SIZE passes and the gfx950 FETCH correction) x the units of this run's launch."""
    k = pmc.get('kernels', {}).get(kernel)
    return k['hbm_bytes_per_unit'] * units if k and units else None


def rooflines(timed, pmc, fps_bytes_per_launch, units, lib_version=None):
    """`roofline` (dominant kernel) + `roofline_other` (every other kernel with a roofline class in SURVEY 8d)."""
    main = roof_entry(timed, 'cyl_net', 'k_cyl_net_wg (A11 Cylindrical_Net, fused fp32 MFMA, Winograd F(2x2,3x3))', 'mfma', MFMA_F32_PEAK_TFLOPS,
                      'TFLOP/s', 1e12, traffic_of(pmc, 'k_cyl_net_wg', units.get('patches')),
                      useful_fraction_of_dense=CYL_NET_EXECUTED_FRACTION)
    if main:
        # The kernel evaluates the reference's convolutions in the Winograd F(2x2,3x3) domain in fp32: 40 MFMAs of 16x16x4 per (4 input,
        # 16 output channels) instead of the 9 x 140 x 4 x 16 MACs of the direct form.  achieved / frac = USEFUL Winograd-domain
        # flops (40 x 1024 MACs per block; the half-empty bottom-row M-tile is inside that count: issued = useful, 8 of the 40 MFMAs
        # carry 8 of 16 rows); the dense algorithmic count of SURVEY 8d (0.1187 GFLOP/patch) over the same time rides along.
        dense = main['achieved']
        main['dense_equivalent_tflops'] = dense
        main['achieved'] = dense * CYL_NET_EXECUTED_FRACTION
        main['frac'] = main['achieved'] / MFMA_F32_PEAK_TFLOPS
        main['avg_issued_flops'] = main['avg_algorithmic_flops'] * CYL_NET_EXECUTED_FRACTION
        main['flops'] = 'achieved = flops of the ISSUED v_mfma_f32_16x16x4_f32 (0.508 of the dense count) / HIP-event time'
        main['traffic_source'] = 'replayed from profiles/traffic.json (PMC passes of this build) x patches per launch' if main['traffic'] else None
    matches = timed['cost_net'][2] / max(timed['cost_net'][0], 1) / COST_NET_FLOPS_PER_MATCH
    other = [
        roof_entry(timed, 'cost_net', 'k_cost_net (A13 CostVolume + CostNet, fused fp32 MFMA)', 'mfma', MFMA_F32_PEAK_TFLOPS, 'TFLOP/s', 1e12,
                   traffic_of(pmc, 'k_cost_net', matches),
                   flops='useful-tile count (0.0519 GFLOP/match: layer 0 separated, layers 1..5 Winograd); dense count of SURVEY 8d: 0.160',
                   dense_equivalent_tflops=((COST_NET_DENSE_FLOPS_PER_MATCH * matches) / (timed['cost_net'][1] / timed['cost_net'][0] * 1e-3) / 1e12
                                            if timed['cost_net'][0] else None)),
        roof_entry(timed, 'grid_query', 'k_grid_query_cell + k_grid_query_wave (A2 radius neighbours: cell-centric self queries, query-centric others)', 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9,
                   traffic_of(pmc, 'k_grid_query', units.get('pairs'))),
        roof_entry(timed, 'grid_subsample', 'k_vox_fused + k_vox_concat (A1 grid subsample, whole call: one workgroup per element, sorted in LDS)', 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9),
        roof_entry(timed, 'vn_gather', 'k_vn_gather (A4 fused VN neighbour block)', 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9,
                   traffic_of(pmc, 'k_vn_gather', units.get('pairs'))),
        roof_entry(timed, 'select_patches', 'k_select_patches_grid (A8 ball query + grouping)', 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9,
                   traffic_of(pmc, 'k_select_patches_grid', units.get('patches_per_select'))),
        voxelize_entry(timed, traffic_of(pmc, 'k_patch_voxelize', units.get('patches')), units.get('patches'), lib_version),
        roof_entry(timed, 'desc_head', 'k_desc_head (A11 attention pooling + normalisation)', 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9,
                   traffic_of(pmc, 'k_desc_head', units.get('patches'))),
        nn1_entry(timed),
    ]
    n, ms, rounds = timed['fps']
    if n:
        by = fps_bytes_per_launch
        other.append({'kernel': 'k_fps (A6 furthest point sampling)', 'bound': 'latency', 'launches': n, 'avg_us': ms / n * 1e3,
                      'rounds_per_launch': rounds / n, 'us_per_round': ms / rounds * 1e3,
                      'achieved': by / (ms / n * 1e-3) / 1e9 if by else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                      'frac': by / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS if by else None, 'traffic': None,
                      'avg_algorithmic_bytes': by})
    return main, [e for e in other if e]


def roofline_split(timed, pmc, units, err):
    """`roofline_split`: csrc/convnet_h3.hip in the split region (achieved = flops of the ISSUED f16 matrix instructions)."""
    n, ms, work = timed['cyl_net_split']
    if not n:
        return None
    patches = work / n / CYL_NET_DENSE_FLOPS_PER_PATCH
    sec = ms / n * 1e-3
    issued = CYL_NET_SPLIT_ISSUED_FLOPS_PER_PATCH * patches
    e = {'kernel': 'k_cyl_net_h3 (A11 Cylindrical_Net, direct 9-tap form on the f16 matrix pipe)',
         'arithmetic': "f16x3 split (x = hi + 2^-11 lo', 3 v_mfma_f32_16x16x32_f16 into 2 fp32 accumulators), fp32 accumulate, fp32-equivalent",
         'bound': 'mfma', 'achieved': issued / sec / 1e12, 'peak': MFMA_F16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
         'frac': issued / sec / 1e12 / MFMA_F16_PEAK_TFLOPS, 'traffic': traffic_of(pmc, 'k_cyl_net_h3', patches),
         'launches': n, 'avg_us': ms / n * 1e3, 'avg_issued_flops': issued, 'avg_algorithmic_flops': work / n,
         'dense_equivalent_tflops': work / n / sec / 1e12,
         'flops': f'achieved = {cyl_net_split_mfmas_per_patch()} issued MFMAs per patch (layer shapes) x 16384 flops / HIP-event time; dense algorithmic count 0.1187 GFLOP/patch'}
    e.update(err or {})
    nc, msc, wc = timed['cost_net_split']
    if nc:                                           # csrc/costnet_h3.hip rides along: dense-equivalent rate only (work = SURVEY 8d's 0.160 GFLOP/match)
        e['cost_net_split'] = {'kernel': 'k_cost_net_h3', 'avg_us': msc / nc * 1e3, 'dense_equivalent_tflops': wc / nc / (msc / nc * 1e-3) / 1e12,
                               'matches_per_launch': wc / nc / COST_NET_DENSE_FLOPS_PER_MATCH}
    return e


def compact(e, algo_key=None):
    """one roofline_other entry reduced to what the driver's 2000-character tail must keep"""
    tr = None
    by = e.get('avg_algorithmic_bytes')
    if e.get('traffic') and by:
        tr = round(e['traffic'] / by, 2)
    out = {'kernel': e['kernel'].split(' ')[0], 'bound': e['bound'], 'frac': None if e.get('frac') is None else round(e['frac'], 4),
           'avg_us': round(e['avg_us'], 1), 'traffic_ratio': tr}
    if 'traffic_ratio_vs_algorithmic_plus_pf' in e:      # k_vn_gather: against algorithmic + the hoisted form's PF table (see main())
        out['traffic_ratio_incl_pf'] = round(e['traffic_ratio_vs_algorithmic_plus_pf'], 2)
    return out


def dgr_ok(poses, gts, rre_deg=15.0):
    ok = 0
    for pose, gt in zip(poses, gts):
        T = np.asarray(pose, np.float64)
        rte = np.linalg.norm(T[:3, 3] - gt[:3, 3])
        rre = np.degrees(np.arccos(np.clip((np.trace(T[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1, 1)))
        ok += int(rte < 0.3 and rre < rre_deg)
    return ok


# ------------------------------------------------------------------------------------------------ main
def _make_sample(job):
    kind, seed = job
    from buffer_amd import synth
    return (synth.make_kitti_pair if kind == 'kitti' else synth.make_pair)(seed)


def make_samples(kind, seeds):
    """The synthetic pairs of this rank (numpy only), generated by a pool of forked workers BEFORE anything touches the GPU
    (33 pairs took ~35 s of the untimed set-up when made one after the other)."""
    import multiprocessing as mp
    jobs = [(kind, s) for s in seeds]
    workers = max(1, min(len(jobs), len(os.sched_getaffinity(0)), 32))      # (N > 1: the affinity is this rank's share, dist.pin_rank)
    # Under a profiler (rocprofv3 preloads its tool library, which initialises the GPU runtime before main) a forked child inherits a
    # half-initialised runtime and its signal handlers: a worker can hang in the tool's finaliser when the pool is torn down (one
    # profiled run of round 4 sat there for an hour).  No forks then: the samples are made one after the other.
    profiled = any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '') \
        or bool(os.environ.get('BUF_NO_TRAFFIC')) or bool(os.environ.get('BUF_BENCH_NO_FORK'))
    if workers == 1 or profiled:
        return [_make_sample(j) for j in jobs]
    pool = mp.get_context('fork').Pool(workers)
    try:
        out = pool.map(_make_sample, jobs)
        pool.close()                                   # workers leave through their normal exit, not through SIGTERM
        pool.join()
        return out
    except BaseException:
        pool.terminate()
        raise


def cnn_error_vs_float64(pipe_f32, pipe_split, dev):
    """max |y - y64| / max |y64| of both descriptor-CNN kernels on 64 post-ReLU-like random patches, the stack in float64 (torch) as truth"""
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.relu(torch.randn((64, 48, 140), generator=g)).to(dev)
    h = x.double().reshape(-1, 48, 7, 20)
    for w, b, relu in pipe_f32.desc.layers:
        h = torch.cat([h[..., -1:], h, h[..., :1]], -1)
        h = torch.nn.functional.pad(h, (0, 0, 1, 1))
        h = torch.nn.functional.conv2d(h, torch.from_numpy(w).double().to(dev), torch.from_numpy(b).double().to(dev))
        h = torch.relu(h) if relu else h
    sc = h.abs().max().item()
    return {'error_vs_float64': (pipe_split.desc.fused(x).double() - h).abs().max().item() / sc,
            'error_vs_float64_fp32_kernel': (pipe_f32.desc.fused(x).double() - h).abs().max().item() / sc}


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
    t_setup = time.perf_counter()
    from buffer_amd.dist import pin_rank
    pinned = pin_rank()                               # N > 1: this rank's host thread and sample pool on its GPU's NUMA-local cores (before any GPU call)
    kitti = a.workload == 'kitti'
    keypts = a.keypts or (1500 if kitti else 5000)
    strong = a.total_pairs is not None
    if strong:                                        # STRONG scaling: the job's pairs of a step are dealt i -> rank i mod N
        if a.workload != 'pair':
            raise SystemExit('--total-pairs applies to --workload pair')
        pps = len(range(rank, a.total_pairs, world))
        job_pairs_per_step = a.total_pairs
    else:
        pps = a.pairs_per_step or (16 if kitti else 32)
        job_pairs_per_step = world * pps
    samples = None
    if a.workload not in ('stream', 'surface'):                        # host-side synthetic data first: forked workers, no GPU state yet
        n_distinct = max(a.distinct_pairs or pps, 1)
        seeds = [1000] + [2000 + rank * 1000 + i for i in range(n_distinct)]
        made = make_samples('kitti' if kitti else 'pair', seeds)
        calib, samples = made[0], made[1:]
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the product has no CPU path)')
    local = local % torch.cuda.device_count()      # (only matters for the single-GPU gloo run of the N>1 logic, see tests)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    backend = os.environ.get('BENCH_BACKEND', 'nccl')   # 'gloo' = the N>1 code path on one GPU (tests/test_bench_contract_gpu.py)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == 'nccl' else torch.device('cpu')

    from buffer_amd import _lib
    from buffer_amd.config import KITTI, THREEDMATCH
    from buffer_amd.pipeline import BufferPipeline
    L = _lib.lib()
    if a.workload == 'stream':
        return run_stream(a, rank, world, dev, cdev, dist, L)
    if a.workload == 'surface':
        if world != 1:
            raise SystemExit('--workload surface measures one caller on one GPU: run it with --gpus 1')
        return run_surface(a, dev, L)
    cfg = replace(KITTI if kitti else THREEDMATCH, num_keypts=keypts, cnn_arith=a.arith)
    pipe = BufferPipeline(cfg, dev)
    limits = pipe.calibrate([calib])                  # same calibration pair on every rank -> identical limits
    inputs = [pipe.upload(s) for s in samples]
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup

    # optional: the pairs of a step split over several stacked batches, one host thread + HIP stream each
    import threading
    from concurrent.futures import ThreadPoolExecutor
    nconc = max(1, min(a.streams, max(pps, 1)))
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

    pool = ThreadPoolExecutor(max_workers=nconc) if nconc > 1 else None

    def step(pp, i):
        ks = [(i * pps + j) % len(inputs) for j in range(pps)]
        if not ks:
            return []
        if pool is None:
            return pp.register_batch([inputs[k] for k in ks], seeds=ks)

        def one_batch(kk):
            with torch.cuda.stream(_bind()):
                return pp.register_batch([inputs[k] for k in kk], seeds=kk)
        parts = [ks[j::nconc] for j in range(nconc)]
        outs = list(pool.map(one_batch, parts))
        poses = [None] * len(ks)
        for j, o in enumerate(outs):
            poses[j::nconc] = o
        return poses

    def run_steps(pp, first, count):
        """`count` steps -> list of poses.  Default: the steps are software-pipelined over two HIP streams
        (BufferPipeline.register_batches: keypoint stage of step i+1 beside the CNN kernels of step i)."""
        if pps == 0:
            return []
        if pool is None and not a.no_pipeline:
            ks = [[((first + i) * pps + j) % len(inputs) for j in range(pps)] for i in range(count)]
            return [p for ps in pp.register_batches([[inputs[k] for k in kk] for kk in ks], seeds=ks) for p in ps]
        return [p for i in range(count) for p in step(pp, first + i)]

    per_rank = []

    def timed_region(pp, warmup):
        """warm-up, then EXACTLY a.steps steps between barrier + synchronize on both sides; max over ranks."""
        run_steps(pp, 0, warmup)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        L.buf_timing_enable(1)
        t0 = time.perf_counter()
        w0 = pp.host_wait_s
        poses = run_steps(pp, a.warmup, a.steps)
        # host share = wall time of the enqueueing loop minus what it spent blocked in the path's two host round trips (candidate
        # counts, match counts); the HIP runtime spin-waits, so CPU-time clocks would read 100 % here
        host_cpu = (time.perf_counter() - t0) - (pp.host_wait_s - w0)
        mine = (torch.stack(poses) if poses else torch.zeros((0, 4, 4), device=dev)).to(cdev)
        gathered = None
        if dist:                                           # the path's one exchange: poses of every shard
            if strong:                                     # shard sizes differ by one: pad to the largest
                cap = a.steps * len(range(0, a.total_pairs, world))
                padded = torch.zeros((cap, 4, 4), dtype=mine.dtype, device=cdev)
                padded[:mine.shape[0]] = mine
                gathered = [torch.empty_like(padded) for _ in range(world)]
                dist.all_gather(gathered, padded)
            else:
                gathered = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(gathered, mine)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        L.buf_timing_enable(0)
        timed = collect_timed(L)
        # first-contact diagnostics of a multi-GPU run: every rank's own clock, host-thread CPU share, set-up time and dominant-kernel
        # time ride in ONE small all_gather (which also yields the max-over-ranks time: no separate all_reduce)
        n_c, ms_c, _ = timed['cyl_net_split' if pp.cfg.cnn_arith == 'split' else 'cyl_net']
        row = [elapsed, host_cpu / max(elapsed, 1e-9), setup_s, ms_c / max(n_c, 1), float(pinned.get('cpus', 0)),
               float(-1 if pinned.get('numa_node') is None else pinned['numa_node'])]
        rows = [row]
        if dist:
            t = torch.tensor(row, dtype=torch.float64, device=cdev)
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            rows = [[float(x) for x in p.cpu().tolist()] for p in parts]
            elapsed = max(r[0] for r in rows)
        per_rank.clear()
        per_rank.extend({'rank': i, 'elapsed_s': round(r[0], 4), 'host_busy_frac': round(r[1], 3), 'setup_s': round(r[2], 1),
                         'main_kernel_ms': round(r[3], 2), 'pinned_cpus': int(r[4]), 'numa_node': None if r[5] < 0 else int(r[5])}
                        for i, r in enumerate(rows))
        return poses, mine, gathered, elapsed, timed

    all_poses, mine, gathered, elapsed, timed = timed_region(pipe, a.warmup)
    per_rank_main = list(per_rank)
    # Kernel characterisation pass (NOT timed, rank 0): two steps one after the other on one stream, so that every kernel of
    # `roofline_other` is measured alone on the chip.  In the pipelined timed region the short keypoint-stage kernels of step i+1
    # share the chip with the CNN kernels of step i: their event spans there measure the contention, not the kernel.
    timed_alone = timed
    latency = {}
    pf_bytes = None
    if rank == 0 and pool is None and not a.no_pipeline and pps:
        from buffer_amd import ops as _ops
        _ops.PF_BYTES[0] = 0.0
        L.buf_timing_enable(1)
        for i in range(2):
            step(pipe, a.warmup + a.steps + i)
        torch.cuda.synchronize()
        L.buf_timing_enable(0)
        timed_alone = collect_timed(L)
        pf_bytes = _ops.PF_BYTES[0]                   # bytes of the VN gather's hoisted-form table (written + re-read) over those two steps
    if rank == 0 and pps and not os.environ.get('BUF_NO_TRAFFIC'):      # (not in the PMC-profiled runs: full-step launches only there)
        # what ONE caller of models/BUFFER.py:231-333 sees: one pair, un-batched, un-pipelined, host clock around a synchronised
        # call (median of 10); and the keypoint stage alone (pyramid, point learner, FPS), for one pair and for a whole step
        def med(fn, n=10):
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts))
        pipe.register_batch([inputs[0]], seeds=[0])
        latency['single_pair_latency_ms'] = med(lambda: pipe.register_batch([inputs[0]], seeds=[0]))
        latency['keypoint_stage_ms'] = {'one_pair': med(lambda: pipe._keypoints([inputs[0]], [0], None)),
                                        'per_step': med(lambda: pipe._keypoints([inputs[k % len(inputs)] for k in range(pps)], list(range(pps)), None), 3)}
        if not kitti:
            # the same single caller at the REFERENCE's operating point (ThreeDMatch/config.py:48: 1500 keypoints per fragment), both arithmetics
            lat15 = {}
            for ar in ('f32', 'split'):
                p15 = BufferPipeline(replace(cfg, num_keypts=1500, cnn_arith=ar), dev, limits=limits)
                p15.register_batch([inputs[0]], seeds=[0])
                lat15[ar] = round(med(lambda: p15.register_batch([inputs[0]], seeds=[0])), 3)
                del p15
            latency['single_pair_latency_ms_1500'] = lat15
    gts = [samples[(a.warmup * pps + n) % len(samples)]['relt_pose'] for n in range(len(all_poses))]    # step i, slot j -> pair (i*pps + j) mod distinct
    ok = dgr_ok(mine.cpu().numpy(), gts)
    ok_ref = dgr_ok(mine.cpu().numpy(), gts, 1.0) if kitti else None      # KITTI/test.py:66-72 as coded: RTE < 0.3 m and RRE < 1 deg

    # second timed region: the SAME steps with the opt-in fp32-equivalent split-f16 CNN kernels (cnn_arith='split')
    split = None
    if not a.no_split and a.arith == 'f32':
        pipe_s = BufferPipeline(replace(cfg, cnn_arith='split'), dev, limits=limits)
        poses_s, mine_s, _, elapsed_s, timed_s = timed_region(pipe_s, 1)
        pipe_s.desc.fused.check_range()
        pipe_s.inlier.fused.check_range()
        lat_s = None
        if rank == 0 and pps and not os.environ.get('BUF_NO_TRAFFIC'):
            ts = []
            for _ in range(11):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pipe_s.register_batch([inputs[0]], seeds=[0])
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            lat_s = float(np.median(ts[1:]))
        if rank == 0:
            dpose = float((mine_s - mine).abs().max().item()) if mine.numel() else 0.0
            # (the 64-patch float64 probe stays out of the PMC-profiled runs: full-step launches only there)
            err = None if os.environ.get('BUF_NO_TRAFFIC') else cnn_error_vs_float64(pipe, pipe_s, dev)
            split = dict(elapsed=elapsed_s, timed=timed_s, ok=dgr_ok(mine_s.cpu().numpy(), gts), dpose=dpose, latency=lat_s, err=err)

    if rank == 0:
        pairs = job_pairs_per_step * a.steps
        per_launch = max(pps, 1) / nconc                                  # pairs covered by one stacked launch
        pmc = load_traffic()
        n_sel = timed['select_patches'][0]
        units = {'pairs': per_launch, 'patches': 2 * keypts * per_launch,
                 'patches_per_select': 2 * keypts * pps * a.steps / n_sel if n_sel else None}
        # A6 bytes: 12 N' + 4 P per cloud; N' (points above the score threshold) <= the sds cloud sizes
        fps_bytes = (sum(12.0 * int(x) for inp in inputs for x in inp['lengths']) / len(inputs) + 8.0 * keypts) * per_launch
        main_roof, _ = rooflines(timed, pmc, fps_bytes, units)                 # dominant kernel: events of the timed region
        if a.arith == 'split':
            main_roof = roofline_split(timed, pmc, units, None)
        units['patches_per_select'] = 2 * keypts * per_launch
        _, other = rooflines(timed_alone, pmc, fps_bytes, units, int(L.buf_version()))     # the others: each kernel alone on the chip
        for e in other:
            # A4: the hoisted form of the four resnet blocks writes PF = [Wf f | Wd f] once per support row and reads it back (a deliberate trade
            # that halved the kernel's time, csrc/vn.hip): its traffic is judged against algorithmic + PF bytes
            if e['kernel'].startswith('k_vn_gather') and pf_bytes and e.get('traffic') and e.get('launches'):
                e['pf_bytes_per_launch'] = pf_bytes / e['launches']
                e['traffic_ratio_vs_algorithmic'] = e['traffic'] / e['avg_algorithmic_bytes']
                e['traffic_ratio_vs_algorithmic_plus_pf'] = e['traffic'] / (e['avg_algorithmic_bytes'] + e['pf_bytes_per_launch'])
        alone_steps = 2 if timed_alone is not timed else a.steps
        label = ('KITTI-shape scan pair (~120k returns per scan, 0.05 / 0.30 m voxels, KITTI constants), full BUFFER inference '
                 '(BASELINE configs[3])') if kitti else 'one 3DMatch-shape fragment pair, full BUFFER inference (BASELINE configs[1])'
        out = {
            'metric': 'registration pairs/sec', 'value': pairs / elapsed, 'unit': 'pairs/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None, 'dtype': 'f32' if a.arith == 'f32' else 'f32 (split-f16 matrix products, fp32 accumulate)', 'data': 'synthetic',
            'config': {'workload': label, 'pairs_per_step_per_gpu': pps, 'pairs_per_step_job': job_pairs_per_step, 'streams': nconc,
                       'steps_pipelined': bool(pool is None and not a.no_pipeline), 'keypoints_per_fragment': keypts,
                       'distinct_pairs_per_gpu': len(samples),
                       'fds_points': [int(samples[0]['src_fds_pts'].shape[0]), int(samples[0]['tgt_fds_pts'].shape[0])],
                       'sds_points': [int(x) for x in inputs[0]['lengths']], 'neighbor_limits': limits,
                       'weights': ('KITTI 06050001' if kitti else '3DMatch 06132318') + ' (released)', 'parallelism': f'pair-sharded x{world}',
                       'registered_ok': f'{ok}/{len(all_poses)} (rank 0, RTE<0.3 m & RRE<15 deg)',
                       **({'registered_ok_reference_criterion': f'{ok_ref}/{len(all_poses)} (KITTI/test.py:66-72: RTE<0.3 m & RRE<1 deg)'} if kitti else {}),
                       'gathered_poses': [int(g.shape[0]) for g in gathered] if gathered else None,
                       'setup_s': round(setup_s, 1), 'per_rank': per_rank_main},
        }
        if world == 1 and not a.no_cpu_baseline:
            def hip_register(perms, seed):
                return pipe.register(inputs[0], seed=seed, perms=[torch.from_numpy(p).to(dev) for p in perms], detail=True)
            hip_register_split = None
            if split:
                def hip_register_split(perms, seed):
                    return pipe_s.register(inputs[0], seed=seed, perms=[torch.from_numpy(p).to(dev) for p in perms], detail=True)
            out['cpu_baseline'] = cpu_baseline(samples[0], cfg, limits, hip_register, hip_register_split)
        out['roofline'] = main_roof
        detail = dict(out, roofline_other=other, timed_kernel_ms_per_step={k: v[1] / alone_steps for k, v in timed_alone.items() if v[0]})
        if split:
            out['roofline_split'] = roofline_split(split['timed'], pmc, units, split['err'])
            out['value_split'] = pairs / split['elapsed']
            out['ms_per_step_split'] = split['elapsed'] / a.steps * 1e3
            out['split'] = {'registered_ok': f"{split['ok']}/{len(all_poses)}", 'max_abs_pose_difference_vs_f32_kernels': split['dpose'],
                            'single_pair_latency_ms': split['latency'],
                            'headline': "value / dtype stay on the fp32-MFMA kernels; *_split = the same steps with cnn_arith='split'"}
            detail.update({k: out[k] for k in ('roofline_split', 'value_split', 'ms_per_step_split', 'split')})
        # compact tail (the driver keeps the last 2000 characters of the line): every other kernel as {kernel, bound, frac, avg_us,
        # traffic_ratio = PMC HBM bytes / algorithmic bytes}, measured alone on the chip in 2 un-pipelined steps after the timed region
        out['roofline_other'] = [compact(e) for e in other]
        out.update(latency)
        detail.update(latency)
        if a.detail_json:
            os.makedirs(os.path.dirname(os.path.abspath(a.detail_json)), exist_ok=True)
            with open(a.detail_json, 'w') as f:
                json.dump(detail, f)
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ drop-in surface
def surface_cpu_leg(calls):
    """The CPU leg of --workload surface (baseline only; the one place of this workload that touches oracle/): every recorded call
    again on the host cores -- the reference's own cpp_wrappers cores (oracle/_ref, compiled from /root/reference in the build
    container) for batch_query / subsample_batch, the plain-C restatements of the third-party CUDA operators for the rest (kind
    'port'; scalar, one thread), numpy for the two gathers and the SVD.  Also the checker: each CPU result is compared with the shim's."""
    from oracle import cpu
    cpu.build(ref=True)
    use_ref = cpu.have_ref()
    rq = cpu.ref_radius_neighbors if use_ref else cpu.radius_neighbors
    rs = cpu.ref_grid_subsample_batch if use_ref else cpu.grid_subsample_batch

    def rows_sorted(x):
        return x[np.lexsort(x.T[::-1])]

    impl = {
        'batch_query': lambda q, s, qb, sb, radius: rq(q, s, qb, sb, radius),
        'subsample_batch': lambda p, b, sampleDl: rs(p, b, sampleDl),
        'furthest_point_sample': lambda xyz, m: cpu.fps(xyz, m),
        'gather_operation': cpu.gather_operation,
        'ball_query': lambda r, ns, xyz, new: cpu.ball_query(r, ns, xyz, new),
        'grouping_operation': cpu.grouping_operation,
        'KNN': lambda ref, query: cpu.knn(ref, query, 1),
        'svd': lambda m: np.linalg.svd(m.astype(np.float32)),
        'three_nn': lambda unknown, known: cpu.three_nn(unknown, known),
    }
    for c in calls:
        fn = impl[c['fn']]
        t0 = time.perf_counter()
        want = fn(*c['host_args'])
        c['ms_cpu'] = (time.perf_counter() - t0) * 1e3
        got = c['host_out']
        if c['fn'] == 'batch_query':              # index for index except inside groups of exactly equidistant neighbours (INTEGRATION.md caveat 1)
            w = min(got.shape[1], want.shape[1])
            c['agree'] = float((got[:, :w] == want[:, :w]).mean()) if got.shape[0] == want.shape[0] else 0.0
            c['equal'] = bool(got.shape == want.shape and c['agree'] > 0.995)
        elif c['fn'] == 'subsample_batch':        # same multiset of rows bit for bit; the reference's row order is its unordered_map's
            c['equal'] = bool(np.array_equal(got[1], want[1]) and got[0].shape == want[0].shape
                              and all(np.array_equal(rows_sorted(got[0][lo:hi]), rows_sorted(want[0][lo:hi]))
                                      for lo, hi in zip(np.cumsum(got[1]) - got[1], np.cumsum(got[1]))))
        elif c['fn'] == 'KNN':
            c['equal'] = bool(np.array_equal(got[1], want[1]) and np.allclose(got[0], want[0], rtol=1e-6, atol=1e-7))
        elif c['fn'] == 'three_nn':               # (dist f32[B,n,3], idx int32[B,n,3])
            c['equal'] = bool(np.array_equal(got[1], want[1]) and np.allclose(got[0], want[0], rtol=1e-6, atol=1e-7))
        elif c['fn'] == 'svd':                    # signs are free: singular values, and U S V^T = A
            u, sv, v = got
            rec = np.einsum('bij,bj,bkj->bik', u, sv, v)
            sc = max(float(np.abs(c['host_args'][0]).max()), 1e-30)
            c['equal'] = bool(np.abs(sv - want[1]).max() < 2e-5 * sc and np.abs(rec - c['host_args'][0]).max() < 2e-5 * sc)
        else:
            c['equal'] = bool(np.array_equal(got, want))
    return 'reference cores (cpp_wrappers) + plain-C port (third-party operators)' if use_ref else 'port'


def run_surface(a, dev, L):
    """--workload surface: what a maintainer gets who ONLY swaps the imports.  One 3DMatch-shape pair at the
    reference's own operating point (ThreeDMatch/config.py:48: 1500 keypoints); every call the reference's inference makes into a
    replaced package, in its order, shapes and host conventions, through buffer_amd/shims:
      ThreeDMatch/dataloader.py:155-224   7 x cpp_neighbors.batch_query + 2 x cpp_subsampling.subsample_batch, numpy in / numpy out
      models/BUFFER.py:266-271            2 x furthest_point_sample ([1,N',3] -> 1500) + 4 x gather_operation
      models/patch_embedder.py:100-104    per cloud ball_query(0.3, 512) + grouping_operation over the shuffled 2 cm cloud
      utils/common.py:442-455             per cloud ball_query(delta / rad_n, 10) + grouping_operation, batch = the 1500 patches, 420 centres
      models/BUFFER.py:347,352            2 x KNN(k=1, transpose_mode=True) on [1,1500,32]
      utils/common.py:715                 svd of [1500,3,3] covariances (cal_Z_axis: only when no reference axis is passed -- off the inference path)
    Inputs of every call are the real intermediates of this pair (BufferPipeline.register(detail=True), untimed).  Each call is timed
    as its caller sees it: host clock around the call, device synchronised on both sides, median of `--steps` repetitions.  Beside
    each: the same call on the host cores (surface_cpu_leg) and, per stage, the fused device form BufferPipeline uses instead."""
    from buffer_amd import ops, pyramid, shims
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.pipeline import BufferPipeline
    from buffer_amd.point_learner import orient_axes
    shims.install()
    import cpp_wrappers.cpp_neighbors.radius_neighbors as cpp_neighbors
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
    import pointnet2_ops.pointnet2_utils as pnt2
    from knn_cuda import KNN
    from torch_batch_svd import svd

    reps = max(a.steps, 3)
    keypts = a.keypts or 1500
    cfg = replace(THREEDMATCH, num_keypts=keypts, cnn_arith=a.arith)
    calib, sample = _make_sample(('pair', 1000)), _make_sample(('pair', 2000))
    pipe = BufferPipeline(cfg, dev)
    limits = pipe.calibrate([calib])
    inp = pipe.upload(sample)
    rng = np.random.default_rng(0)
    perms = [torch.from_numpy(rng.permutation(int(r.shape[0]))).to(dev) for r in (inp['src_raw'], inp['tgt_raw'])]
    _, d = pipe.register(inp, seed=0, perms=perms, detail=True)
    torch.cuda.synchronize()

    calls = []

    def host(x):
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
        if isinstance(x, (tuple, list)):
            return tuple(host(y) for y in x)
        return x

    def call(op, fn_name, site, shape, fn, *args, **kw):
        """one call of the surface: warm once, then the median of `reps` synchronised repetitions on the host clock"""
        out = fn(*args, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*args, **kw)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        calls.append(dict(op=op, fn=fn_name, site=site, shape=shape, ms_gpu=float(np.median(ts)),
                          host_args=tuple(host(x) for x in args) + tuple(host(v) for v in kw.values()), host_out=host(out)))
        return out

    # ---- collate (ThreeDMatch/dataloader.py:155-224): numpy in, numpy out, as the DataLoader worker calls it
    pts = np.concatenate([sample['src_sds_pts'][:, :3], sample['tgt_sds_pts'][:, :3]]).astype(np.float32)
    lens = np.array([len(sample['src_sds_pts']), len(sample['tgt_sds_pts'])], np.int32)
    r = cfg.voxel_size_0 * cfg.conv_radius
    bq, sb = 'cpp_neighbors.batch_query', 'cpp_subsampling.subsample_batch'
    for layer in range(3):
        call(bq, 'batch_query', 'ThreeDMatch/dataloader.py:172', f'conv l{layer}: {len(pts)} x {len(pts)}, r={r:.2f}',
             cpp_neighbors.batch_query, pts, pts, lens, lens, radius=r)
        if layer == 2:
            break
        dl = 2 * r / cfg.conv_radius
        pool_p, pool_b = call(sb, 'subsample_batch', 'ThreeDMatch/dataloader.py:190', f'l{layer}: {len(pts)} pts, dl={dl:.3f}',
                              cpp_subsampling.subsample_batch, pts, lens, sampleDl=dl)
        call(bq, 'batch_query', 'ThreeDMatch/dataloader.py:196', f'pool l{layer}: {len(pool_p)} x {len(pts)}, r={r:.2f}',
             cpp_neighbors.batch_query, pool_p, pts, pool_b, lens, radius=r)
        call(bq, 'batch_query', 'ThreeDMatch/dataloader.py:200', f'up l{layer}: {len(pts)} x {len(pool_p)}, r={2 * r:.2f}',
             cpp_neighbors.batch_query, pts, pool_p, lens, pool_b, radius=2 * r)
        pts, lens, r = pool_p, pool_b, 2 * r

    # ---- keypoints (models/BUFFER.py:255-271)
    n_src = int(inp['lengths'][0])
    pts0, score = d['pyr']['points'][0], d['score'][:, 0]
    kp, ka, cand = [], [], []
    for lo, hi in ((0, n_src), (n_src, int(pts0.shape[0]))):
        keep = torch.where(score[lo:hi] > cfg.keypts_th)[0]
        p = pts0[lo:hi][keep].contiguous()
        ax = orient_axes(d['axis'][lo:hi], pts0[lo:hi])[keep].contiguous()
        cand.append(int(p.shape[0]))
        p_f, a_f = p[None].transpose(1, 2).contiguous(), ax[None].transpose(1, 2).contiguous()
        idx = call('pnt2.furthest_point_sample', 'furthest_point_sample', 'models/BUFFER.py:266-267', f'[1,{p.shape[0]},3] -> {keypts}',
                   pnt2.furthest_point_sample, p[None], keypts)
        k1 = call('pnt2.gather_operation', 'gather_operation', 'models/BUFFER.py:268-271', f'[1,3,{p.shape[0]}] by [1,{keypts}]',
                  pnt2.gather_operation, p_f, idx)
        a1 = call('pnt2.gather_operation', 'gather_operation', 'models/BUFFER.py:268-271', f'[1,3,{p.shape[0]}] by [1,{keypts}]',
                  pnt2.gather_operation, a_f, idx)
        kp.append(k1.transpose(1, 2).contiguous())
        ka.append(a1.transpose(1, 2).contiguous())
    assert all(torch.equal(kp[i][0], d['kpts'][i]) for i in range(2)), 'shim keypoints differ from BufferPipeline.register()'

    # ---- patches (models/patch_embedder.py:93-104) and the SPT ball queries (utils/common.py:431-455)
    raws = (inp['src_raw'], inp['tgt_raw'])
    voxel_r = cfg.delta / cfg.rad_n
    centres = pipe.desc.centres[None].repeat(keypts, 1, 1).contiguous()
    covs = []
    for i in range(2):
        cloud = raws[i][perms[i]][None].contiguous()
        nf = int(cloud.shape[1])
        gi = call(f'pnt2.ball_query({cfg.des_r}, {cfg.num_points_per_patch})', 'ball_query', 'models/patch_embedder.py:100',
                  f'[1,{nf},3] x [1,{keypts},3]', pnt2.ball_query, cfg.des_r, cfg.num_points_per_patch, cloud, kp[i])
        call('pnt2.grouping_operation (patches)', 'grouping_operation', 'models/patch_embedder.py:102-104',
             f'[1,3,{nf}] by [1,{keypts},{cfg.num_points_per_patch}]', pnt2.grouping_operation, cloud.transpose(1, 2).contiguous(), gi)
        patches = d['desc'][i]['patches'].contiguous()                      # aligned + normalised [P,512,3]: what SPT receives
        vi = call(f'pnt2.ball_query({voxel_r:.3f}, {cfg.voxel_sample})', 'ball_query', 'utils/common.py:442',
                  f'[{keypts},{patches.shape[1]},3] x [{keypts},{centres.shape[1]},3]', pnt2.ball_query, voxel_r, cfg.voxel_sample, patches, centres)
        call('pnt2.grouping_operation (voxels)', 'grouping_operation', 'utils/common.py:452-455',
             f'[{keypts},3,{patches.shape[1]}] by [{keypts},{centres.shape[1]},{cfg.voxel_sample}]', pnt2.grouping_operation,
             patches.transpose(1, 2).contiguous(), vi)
        covs.append(torch.matmul(patches.transpose(-1, -2), patches))

    cp = [pts0[lo:hi][torch.where(score[lo:hi] > cfg.keypts_th)[0]].contiguous() for lo, hi in ((0, n_src), (n_src, int(pts0.shape[0])))]
    # ---- mutual matching (models/BUFFER.py:347,352)
    des = [d['desc'][i]['desc'].contiguous() for i in range(2)]
    knn = KNN(k=1, transpose_mode=True)
    call('knn_cuda.KNN(k=1)', 'KNN', 'models/BUFFER.py:347', f'ref [1,{keypts},32], query [1,{keypts},32]', knn, des[1][None], des[0][None])
    call('knn_cuda.KNN(k=1)', 'KNN', 'models/BUFFER.py:352', f'ref [1,{keypts},32], query [1,{keypts},32]', knn, des[0][None], des[1][None])
    for i in range(2):
        call('torch_batch_svd.svd', 'svd', 'utils/common.py:715 (cal_Z_axis: off the inference path, z_axis is given)', f'[{keypts},3,3]', svd, covs[i])
    for i in range(2):                                                      # the fifth pointnet2 operator of the surface: no call site in the reference's inference
        unknown = cp[i][None].contiguous()
        call('pnt2.three_nn', 'three_nn', 'README.md:31 (pointnet2_ops; no call site on the inference path)', f'[1,{unknown.shape[1]},3] vs [1,{keypts},3]',
             pnt2.three_nn, unknown, kp[i])

    # ---- the fused device forms BufferPipeline runs in place of those calls (same pair, same inputs already in HBM)
    def med(fn):
        fn()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    cat_cand, kp_cat, ka_cat = torch.cat(cp).contiguous(), torch.cat([k[0] for k in kp]).contiguous(), torch.cat([k[0] for k in ka]).contiguous()
    sup = torch.cat([raws[i][perms[i]] for i in range(2)]).contiguous()
    sup_len = [int(raws[0].shape[0]), int(raws[1].shape[0])]
    init_patches = ops.select_patches_batched(sup, sup_len, kp_cat, keypts, cfg.des_r, cfg.num_points_per_patch)
    dd = torch.stack(des)
    fused = [
        dict(stage='pyramid.build_pyramid (3 cell grids, 7 queries, 2 subsamplings; device in, device out)',
             replaces=[bq, sb], ms=med(lambda: pyramid.build_pyramid(inp['points'], inp['lengths'], limits, cfg))),
        dict(stage='ops.furthest_point_sample_ragged (both clouds, one launch) + row gathers',
             replaces=['pnt2.furthest_point_sample', 'pnt2.gather_operation'],
             ms=med(lambda: (lambda f: [cp[i][f[i]] for i in range(2)])(ops.furthest_point_sample_ragged(cat_cand, cand, keypts).long()))),
        dict(stage='ops.select_patches_batched (ball query + grouping + keypoint substitution, both clouds, no index tensor)',
             replaces=[c['op'] for c in calls if c['site'].startswith('models/patch_embedder.py')][:2],
             ms=med(lambda: ops.select_patches_batched(sup, sup_len, kp_cat, keypts, cfg.des_r, cfg.num_points_per_patch))),
        dict(stage='ops.patch_voxelize (axis align + 420 ball queries + var_to_invar + point MLP + max-pool per patch, both clouds)',
             replaces=[c['op'] for c in calls if c['site'].startswith('utils/common.py:4')][:2],
             ms=med(lambda: ops.patch_voxelize(init_patches, ka_cat, cfg.des_r, pipe.desc.centres, pipe.desc.azi_cs, voxel_r, cfg.voxel_sample,
                                               pipe.desc.mlp_w, pipe.desc.mlp_b, pipe.desc.mlp_s, pipe.desc.mlp_t, cfg.azi_n, False))),
        dict(stage='ops.knn x 2 (matrix-pipe ranking, fp32 decision)', replaces=['knn_cuda.KNN(k=1)'],
             ms=med(lambda: (ops.knn(dd[1:2], dd[0:1], 1), ops.knn(dd[0:1], dd[1:2], 1)))),
    ]
    whole = med(lambda: pipe.register_batch([inp], seeds=[0]))

    kind = None
    if not a.no_cpu_baseline:
        kind = surface_cpu_leg(calls)
    table, order = {}, []
    for c in calls:
        if c['op'] not in table:
            order.append(c['op'])
            table[c['op']] = dict(op=c['op'], site=c['site'], calls=0, ms_gpu=0.0, ms_cpu=0.0 if kind else None,
                                  equal_to_cpu=True if kind else None, shapes=[])
        t = table[c['op']]
        t['calls'] += 1
        t['ms_gpu'] += c['ms_gpu']
        t['shapes'].append(c['shape'])
        if kind:
            t['ms_cpu'] += c['ms_cpu']
            t['equal_to_cpu'] = bool(t['equal_to_cpu'] and c['equal'])
    rows = [table[k] for k in order]
    for t in rows:
        t['ms_gpu'] = round(t['ms_gpu'], 3)
        t['ms_cpu'] = None if t['ms_cpu'] is None else round(t['ms_cpu'], 2)
    on_path = [t for t in rows if 'svd' not in t['op'] and 'three_nn' not in t['op']]
    sum_gpu = sum(t['ms_gpu'] for t in on_path)
    sum_cpu = sum(t['ms_cpu'] for t in on_path) if kind else None
    sum_fused = sum(f['ms'] for f in fused)
    out = {
        'metric': 'registration pairs/sec', 'value': 1e3 / sum_gpu, 'unit': 'pairs/s', 'n_gpus': 1, 'steps': reps, 'warmup': 1,
        'ms_per_step': sum_gpu, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'operator surface of one 3DMatch-shape pair through the shim packages (drop-in mode): the calls the reference '
                               'makes into cpp_wrappers / pointnet2_ops / knn_cuda / torch_batch_svd, in its order, shapes and host conventions; '
                               'value = 1 / (sum of those calls): the surface alone, NOT a registration rate -- the reference\'s torch layers '
                               'between the calls are not part of it',
                   'keypoints_per_fragment': keypts, 'fds_points': sup_len, 'sds_points': [int(x) for x in inp['lengths']],
                   'candidates_above_threshold': cand, 'neighbor_limits': limits, 'repetitions_per_call': reps,
                   'timing': 'host clock around each call, device synchronised on both sides, median; cpp_wrappers calls include their numpy <-> device copies'},
        'surface': {'ops': rows, 'sum_ms_gpu': round(sum_gpu, 3), 'sum_ms_cpu': None if sum_cpu is None else round(sum_cpu, 1),
                    'sum_excludes': 'torch_batch_svd.svd and pnt2.three_nn (no call on the inference path)',
                    'fused': [dict(f, ms=round(f['ms'], 3)) for f in fused], 'sum_ms_fused': round(sum_fused, 3),
                    'buffer_pipeline_whole_pair_ms': round(whole, 3),
                    'buffer_pipeline_note': 'BufferPipeline.register_batch([pair]): the WHOLE inference of this pair (operators + both CNNs + pose recovery), '
                                            'un-batched, host clock'},
    }
    if kind:
        out['cpu_baseline'] = {'value': 1e3 / sum_cpu, 'unit': 'pairs/s', 'cores': 1, 'kind': kind,
                               'sample': 'the same calls with the same inputs, once each, one host thread'}
    if a.detail_json:
        os.makedirs(os.path.dirname(os.path.abspath(a.detail_json)), exist_ok=True)
        with open(a.detail_json, 'w') as f:
            json.dump(dict(out, calls=[{k: v for k, v in c.items() if k not in ('host_args', 'host_out')} for c in calls]), f)
    print(json.dumps(out), flush=True)


def run_stream(a, rank, world, dev, cdev, dist, L):
    """BASELINE configs[2]: `--stream-pairs` synthetic pairs (a quarter at 0.3 overlap) from RAW device-resident clouds:
    voxelisation x2 + normals + registration inside the timed region; pair i -> rank i mod W; one all_gather of poses."""
    from buffer_amd import dist as bdist, stream, synth
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.pipeline import BufferPipeline
    from buffer_amd.threedmatch import upload
    cfg = replace(THREEDMATCH, num_keypts=a.keypts or 1500, cnn_arith=a.arith)
    pipe = BufferPipeline(cfg, dev)
    n = a.stream_pairs
    ids = list(bdist.shard_indices(n, rank, world))
    overlaps = tuple(float(x) for x in a.stream_overlaps.split(',')) if a.stream_overlaps else stream.OVERLAPS
    mine = [synth.make_raw_pair_device(20000 + i, overlaps[i % len(overlaps)], dev) for i in ids]
    first = stream.prepare(synth.make_raw_pair_device(20000, stream.OVERLAPS[0], dev), cfg, 0)     # same pair on every rank
    limits = pipe.calibrate([{k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in first.items()}])
    batch = a.pairs_per_step or 32
    extra = None
    if world > 1:
        # the path's one exchange carries the ground truth of every pair as well (relative pose + the 6 x 6 information matrix of
        # the gt.info proxy: 52 floats, formed here, untimed), so rank 0 scores the whole job without regenerating any other
        # rank's raw clouds
        extra = torch.tensor(np.stack([np.concatenate([np.asarray(m['relt_pose'], np.float64).reshape(-1),
                                                       synth.information_matrix(m['overlap_pts'].cpu().numpy()).reshape(-1)]) for m in mine])
                             if mine else np.zeros((0, 52)), dtype=torch.float32).to(cdev)
    stream.run(pipe, mine[:batch], batch)                                                           # warm-up
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    L.buf_timing_enable(1)
    t0 = time.perf_counter()
    chunks = [list(range(lo, min(lo + batch, len(ids)))) for lo in range(0, len(ids), batch)]
    makers = [(lambda ch=ch: [upload(x) for x in stream.prepare_batch([mine[j] for j in ch], cfg, [ids[j] for j in ch])]) for ch in chunks]
    poses = [p for ps in pipe.register_batches(makers, seeds=[[ids[j] for j in ch] for ch in chunks]) for p in ps]
    local_poses = torch.stack(poses) if poses else torch.zeros((0, 4, 4), device=dev)
    all_poses, gt = local_poses, None
    if world > 1:
        all_poses, gt = bdist.gather_poses(ids, local_poses, n, device=cdev, extra=extra)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L.buf_timing_enable(0)
    timed = collect_timed(L)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        if gt is None:
            meta = [{'relt_pose': m['relt_pose'], 'overlap_pts': m['overlap_pts']} for m in mine]
        else:
            g = gt.cpu().numpy().astype(np.float64)
            meta = [{'relt_pose': g[i, :16].reshape(4, 4), 'info': g[i, 16:].reshape(6, 6)} for i in range(n)]
        quality = stream.evaluate_stream(meta, all_poses.cpu().numpy())
        quality['scored'] = f'{n} pairs (all ranks, gathered poses)'
        main_roof, other = rooflines(timed, load_traffic(), None, {})
        if a.arith == 'split':
            main_roof = roofline_split(timed, load_traffic(), {}, None)
        print(json.dumps({
            'metric': 'registration pairs/sec', 'value': n / elapsed, 'unit': 'pairs/s', 'n_gpus': world, 'steps': 1, 'warmup': 0,
            'ms_per_step': elapsed * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32' if a.arith == 'f32' else 'f32 (split-f16 matrix products, fp32 accumulate)', 'data': 'synthetic',
            'config': {'workload': f'{n} synthetic 3DMatch-shape pairs streamed from raw clouds, pre-processing included '
                                   f'(BASELINE configs[2]); overlaps {list(overlaps)} in equal shares',
                       'keypoints_per_fragment': cfg.num_keypts, 'pairs_per_launch': batch, 'neighbor_limits': limits,
                       'parallelism': f'pair-sharded x{world}'},
            'quality': quality, 'roofline': main_roof, 'roofline_other': [compact(e) for e in other]}), flush=True)
    if dist:
        dist.barrier()                                  # every rank stays until rank 0 has scored and printed
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
