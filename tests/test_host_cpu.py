"""CPU-side checks: C-ABI library surface, shims import, evaluation harness vs the reference's
evaluator (fixture F6), pair sharding over 2 gloo ranks."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    """include/buffer_hip.h <-> libbuffer_hip.so <-> ctypes table (no compute call: no GPU needed)."""
    from buffer_amd import _lib, build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "buffer_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(buf_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == _lib.exported_symbols()
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.buf_version() >= 100
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for name in declared:
        assert re.search(rf"\bT {name}\b", nm), f"{name} is not an exported text symbol"


def test_library_holds_no_packed_fp32_instruction(tmp_path):
    """Round 5: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 return wrong values in 16 lanes now and then while another wavefront of
    the SIMD issues f16 MFMAs (tests/test_concurrency_gpu.py, profiles/r05_packed_fp32_hazard.txt): the build turns them off
    (-target-feature -packed-fp32-ops).  Disassemble the gfx950 code object of the in-tree library and look."""
    import shutil
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('no llvm-objdump')
    from buffer_amd import _lib
    so = str(tmp_path / 'lib.so')
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, '--offloading', so], capture_output=True, cwd=str(tmp_path), check=True)
    co = [f for f in os.listdir(tmp_path) if 'gfx950' in f]
    assert len(co) == 1, os.listdir(tmp_path)
    asm = subprocess.run([objdump, '-d', str(tmp_path / co[0])], capture_output=True, text=True, check=True).stdout
    assert 'v_mfma_f32_16x16x32_f16' in asm and 'k_vn_gather6_lds' in asm            # the right object, disassembled
    packed = re.findall(r'v_pk_(?:mul|add|fma)_f32', asm)
    assert not packed, f'{len(packed)} packed-fp32 instructions in libbuffer_hip.so: build with -target-feature -packed-fp32-ops'


def test_ops_fail_loudly_without_device():
    from buffer_amd import ops, _lib
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.BufferHipError):
        ops.radius_neighbors(torch.zeros(4, 3), torch.zeros(4, 3), [4], [4], 0.1)
    from buffer_amd.pipeline import BufferPipeline
    with pytest.raises(RuntimeError):
        BufferPipeline(device='cpu')


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "buffer_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


def test_shims_resolve_under_reference_names():
    import buffer_amd.shims as shims
    shims.install()
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cs
    import cpp_wrappers.cpp_neighbors.radius_neighbors as cn
    import pointnet2_ops.pointnet2_utils as pnt2
    from knn_cuda import KNN
    from torch_batch_svd import svd
    assert callable(cs.subsample_batch) and callable(cs.subsample) and callable(cn.batch_query)
    for f in ("furthest_point_sample", "gather_operation", "ball_query", "grouping_operation", "three_nn"):
        assert callable(getattr(pnt2, f))
    assert KNN(k=1, transpose_mode=True).k == 1 and callable(svd)
    with pytest.raises(RuntimeError):
        cs.subsample_batch(np.zeros((4, 3)), np.array([4]), method="bogus")
    with pytest.raises(RuntimeError):
        cn.batch_query(np.zeros((4, 2)), np.zeros((4, 3)), [4], [4], radius=0.1)
    with pytest.raises(RuntimeError):
        cn.batch_query(np.zeros((4, 3)), np.zeros((4, 3)), [4], [2, 2], radius=0.1)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            cn.batch_query(np.zeros((4, 3)), np.zeros((4, 3)), [4], [4], radius=0.1)


def test_evaluator_matches_reference_fixture(tmp_path):
    from buffer_amd import evaluate as E
    g = np.load(os.path.join(GOLD, "rr_eval.npz"))
    for k in ("gt_log", "est_log", "gt_info"):
        (tmp_path / k).write_text(str(g[k]))
    gt_pairs, gt_traj = E.read_trajectory(str(tmp_path / "gt_log"))
    n_frag, cov = E.read_trajectory_info(str(tmp_path / "gt_info"))
    est_pairs, est_traj = E.read_trajectory(str(tmp_path / "est_log"))
    assert n_frag == int(g['n_fragments'])
    prec, rec, flags, errs = E.evaluate_registration(n_frag, est_traj, est_pairs, gt_pairs, gt_traj, cov)
    assert np.allclose([prec, rec], g['result'])
    assert np.array_equal(np.asarray(flags), g['flags'])
    np.testing.assert_allclose(errs, g['errors'], rtol=1e-6, equal_nan=True)


def test_log_writer_round_trip(tmp_path):
    from buffer_amd import evaluate as E, synth
    rng = np.random.default_rng(0)
    T = np.eye(4)
    T[:3, :3] = synth.random_rotation(rng)
    T[:3, 3] = rng.normal(size=3)
    path = str(tmp_path / "scene" / "est.log")
    E.append_log(path, 3, 7, T)
    pairs, traj = E.read_trajectory(path)
    assert list(pairs[0][:2]) == ['3', '7']
    np.testing.assert_allclose(traj[0], np.linalg.inv(T), atol=1e-6)       # the log holds the inverse pose
    ok, rte, rre = E.dgr_success(T, T)
    assert ok and rte == 0 and rre < 1e-3


def test_config_constants_and_voxel_centres():
    from buffer_amd.config import THREEDMATCH, KITTI
    from buffer_amd.patch_embedder import voxel_centres
    from oracle import torch_ref as T
    assert THREEDMATCH.hist_n == 34 and THREEDMATCH.scale == 1.0
    assert (KITTI.voxel_size_0, KITTI.des_r, KITTI.keypts_th, KITTI.pose_refine) == (0.30, 3.0, 0.5, False)
    c = voxel_centres(3, 20, 7)
    assert c.shape == (420, 3)
    assert np.array_equal(c, T.voxel_centres(3, 20, 7).numpy())


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from buffer_amd import dist as bd
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
rank, world = dist.get_rank(), dist.get_world_size()
n = 7
ids = bd.shard_indices(n, rank, world)
poses = torch.stack([torch.eye(4) * (i + 1) for i in ids]) if ids else torch.zeros(0, 4, 4)
allp = bd.gather_poses(ids, poses, n)
assert allp.shape == (n, 4, 4)
for i in range(n):
    assert torch.equal(allp[i], torch.eye(4) * (i + 1)), (rank, i)
lim = bd.broadcast_limits([17, 20, 24] if rank == 0 else [0, 0, 0])
assert lim == [17, 20, 24]
dist.destroy_process_group()
print('ok', rank)
'''


def test_pair_sharding_two_gloo_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"ok {r}" in o


_WORKER8 = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from buffer_amd import dist as bd
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
def pose(i):
    T = torch.eye(4); T[0, 3] = float(i); T[1, 3] = -0.5 * i
    return T
# 1623 (3DMatch) and 1781 (3DLoMatch) pairs: neither divides by 8; 5 and 0 pairs: fewer pairs than ranks (empty shards)
for n in (1623, 1781, 5, 0):
    ids = bd.shard_indices(n, rank, world)
    assert len(ids) in (n // world, n // world + 1) and all(i % world == rank for i in ids)
    poses = torch.stack([pose(i) for i in ids]) if ids else torch.zeros(0, 4, 4)
    extra = torch.tensor([[float(i), 2.0 * i, -1.0] for i in ids]).reshape(-1, 3)
    allp, ext = bd.gather_poses(ids, poses, n, extra=extra)
    assert allp.shape == (n, 4, 4) and ext.shape == (n, 3)
    for i in range(n):
        assert torch.equal(allp[i], pose(i)), (rank, n, i)
        assert ext[i].tolist() == [float(i), 2.0 * i, -1.0]
    assert torch.equal(bd.gather_poses(ids, poses, n), allp)                      # without the payload
# any id assignment: the ids travel as int32 in their own exchange (rank r reports the pairs of rank 7 - r, reversed)
n = 29
ids = bd.shard_indices(n, world - 1 - rank, world)[::-1]
poses = torch.stack([pose(i) for i in ids]) if ids else torch.zeros(0, 4, 4)
allp = bd.gather_poses(ids, poses, n, explicit_ids=True)
for i in range(n):
    assert torch.equal(allp[i], pose(i)), (rank, i)
try:
    bd.gather_poses(ids, poses, n)
    raise SystemExit('ids that break the sharding rule were accepted')
except ValueError:
    pass
assert torch.isfinite(allp).all()
dist.destroy_process_group()
print('ok', rank)
'''


def test_pair_sharding_eight_gloo_ranks_uneven_and_empty_shards(tmp_path):
    """SURVEY 8e at the node's world size: 1623 / 1781 pairs (neither divides by 8), fewer pairs than ranks, the ground-truth
    payload in the same exchange, explicit int32 ids; no float block ever carries an id (round 3: id -1 as float bits = NaN)."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="8", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(8)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"ok {r}" in o


def test_rank_cpus_follow_the_numa_node_of_the_rank_s_gpu(tmp_path):
    """dist.rank_cpus on a fake sysfs tree: 8 GPUs, four per socket, 2 x 64 cores -> every rank 16 cores of ITS socket, disjoint,
    all cores used; without NUMA information: contiguous equal shares; more ranks than cores: everything shared."""
    from buffer_amd import dist as bd
    root = tmp_path / 'sys'
    for i in range(8):
        d = root / 'bus' / 'pci' / 'devices' / f'0000:{0x10 + 0x10 * i:02x}:00.0'
        d.mkdir(parents=True)
        (d / 'vendor').write_text('0x1002\n')
        (d / 'class').write_text('0x120000\n')           # processing accelerator
        (d / 'numa_node').write_text(f'{i // 4}\n')
    nic = root / 'bus' / 'pci' / 'devices' / '0000:05:00.0'
    nic.mkdir(parents=True)
    (nic / 'vendor').write_text('0x15b3\n'); (nic / 'class').write_text('0x020000\n'); (nic / 'numa_node').write_text('0\n')
    for n, cl in enumerate(('0-31,64-95', '32-63,96-127')):
        d = root / 'devices' / 'system' / 'node' / f'node{n}'
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(cl + '\n')
    allowed = list(range(128))
    shares = [bd.rank_cpus(r, 8, allowed, str(root)) for r in range(8)]
    assert [n for _, n in shares] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert all(len(c) == 16 for c, _ in shares)
    assert sorted(c for cs, _ in shares for c in cs) == allowed
    node0 = set(bd._parse_cpulist('0-31,64-95'))
    assert all(set(cs) <= node0 for cs, _ in shares[:4]) and all(not (set(cs) & node0) for cs, _ in shares[4:])
    # a cgroup that only allows socket 1: ranks of node 0 have no local core left -> contiguous shares of what is allowed
    cs, node = bd.rank_cpus(0, 8, list(range(32, 64)), str(root))
    assert node is None and cs == [32, 33, 34, 35]
    # no sysfs information at all
    plain = [bd.rank_cpus(r, 4, list(range(10)), str(tmp_path / 'none')) for r in range(4)]
    assert [c for c, _ in plain] == [[0, 1], [2, 3, 4], [5, 6], [7, 8, 9]] and all(n is None for _, n in plain)
    assert bd.rank_cpus(2, 8, [0, 1], str(tmp_path / 'none')) == ([0, 1], None)
    assert bd.pin_rank(0, 1) == {}                        # a single rank owns the node: nothing to pin


def test_shard_indices_cover_everything():
    from buffer_amd import dist as bd
    for n in (0, 1, 7, 1623):
        for w in (1, 2, 8):
            got = sorted(sum((bd.shard_indices(n, r, w) for r in range(w)), []))
            assert got == list(range(n))


def test_mfma_tile_weights_layout():
    """ops.mfma_tile_weights: block (g, n) of 256 floats = [lk][li][p] with K-row 16g + 4p + lk (conv net) or
    16g + 4lk + p (cost net, lk_major) and column 16n + li -- the contract stated in include/buffer_hip.h."""
    import numpy as np
    from buffer_amd.ops import mfma_tile_weights
    K, C = 48, 32
    w = np.arange(K * C, dtype=np.float32).reshape(K, C)
    for lk_major in (False, True):
        t = mfma_tile_weights(w, lk_major=lk_major).reshape(K // 16, C // 16, 4, 16, 4)
        for g, n, lk, li, p in [(0, 0, 0, 0, 0), (2, 1, 3, 15, 2), (1, 0, 2, 7, 3), (2, 1, 0, 9, 1)]:
            k = 16 * g + (4 * lk + p if lk_major else 4 * p + lk)
            assert t[g, n, lk, li, p] == w[k, 16 * n + li]


def test_winograd_filter_tiling_reproduces_the_convolution():
    """ops.winograd_tile_weights (host side of csrc/convnet_wg.hip): un-tiling gives U = G g G^T, and the F(2x2,3x3) identity
    A^T [U (.) B^T d B] A on the 4 x 10 tile grid reproduces the circular-azimuth / zero-elevation 3x3 correlation."""
    from buffer_amd import ops
    rng = np.random.default_rng(3)
    cin, cout = 8, 16
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    x = rng.standard_normal((cin, 7, 20))
    U = ops.winograd_tile_weights(w, ng=1, blocks=5).reshape(cout // 16, 5, cin // 4, 4, 16, 4)    # [n, i, ks, lk, li, j]
    U = np.transpose(U, (1, 5, 0, 4, 2, 3)).reshape(5, 4, cout, cin).astype(np.float64)   # [i, j, Cout, Cin]
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    assert np.abs(U[:4] - np.einsum('ia,ocab,jb->ijoc', G, w.astype(np.float64), G)).max() < 1e-6
    # optional block 4 = the filter's middle row, column-transformed = U_1 - U_2: the second tap of the kernel's bottom-row form
    # (formed in registers from blocks 1 and 2 there)
    assert np.abs(U[4] - np.einsum('ocb,jb->joc', w[:, :, 1].astype(np.float64), G)).max() < 1e-6
    assert np.abs(U[4] - (U[1] - U[2])).max() < 1e-6
    Ub, U = U[4], U[:4]
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    xp = np.zeros((cin, 10, 22))
    xp[:, 1:8, 1:21] = x; xp[:, 1:8, 0] = x[:, :, 19]; xp[:, 1:8, 21] = x[:, :, 0]
    ref = np.zeros((cout, 7, 20))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum('oc,cyx->oyx', w[:, :, ky, kx].astype(np.float64), xp[:, ky:ky + 7, kx:kx + 20])
    out = np.zeros((cout, 8, 20))
    for ty in range(4):
        for tx in range(10):
            V = np.einsum('ia,cab,jb->ijc', BT, xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4], BT)
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum('ui,ijo,vj->ouv', AT, np.einsum('ijoc,ijc->ijo', U, V), AT)
    assert np.abs(out[:, :7] - ref).max() < 1e-5 * np.abs(ref).max()
    # the bottom tile row as the kernel computes it: y[6] = A^T-columns of (colB(d[5]) (.) U_0 + colB(d[6]) (.) block 4)
    for tx in range(10):
        V5, V6 = (np.einsum('cb,jb->jc', xp[:, r, 2 * tx:2 * tx + 4], BT) for r in (6, 7))        # padded rows 6, 7 = map rows 5, 6
        m = np.einsum('joc,jc->jo', U[0], V5) + np.einsum('joc,jc->jo', Ub, V6)
        assert np.abs(np.einsum('vj,jo->ov', AT, m) - ref[:, 6, 2 * tx:2 * tx + 2]).max() < 1e-5 * np.abs(ref).max()
    # the C-ABI host helper produces the same tiling, bit for bit
    import ctypes as C
    from buffer_amd import _lib
    w2 = np.ascontiguousarray(rng.standard_normal((32, 16, 3, 3)).astype(np.float32))
    got = np.empty(16 * 32 * 16, np.float32)
    rc = _lib.lib().buf_winograd_tile_weights(w2.ctypes.data_as(C.c_void_p), 32, 16, got.ctypes.data_as(C.c_void_p))
    assert rc == 0 and np.array_equal(got, ops.winograd_tile_weights(w2))
    assert _lib.lib().buf_winograd_tile_weights(w2.ctypes.data_as(C.c_void_p), 30, 16, got.ctypes.data_as(C.c_void_p)) == -1
    # layers in the paired form (buf_winograd_group == 2): [pair][i][ks][n2][lk][li][j]
    assert _lib.lib().buf_winograd_group(32, 128) == 2 and _lib.lib().buf_winograd_group(48, 64) == 2 and _lib.lib().buf_winograd_group(64, 32) in (1, 2)
    w3 = np.ascontiguousarray(rng.standard_normal((128, 32, 3, 3)).astype(np.float32))
    t3 = ops.winograd_tile_weights(w3)
    U3 = np.transpose(t3.reshape(4, 4, 8, 2, 4, 16, 4), (1, 6, 0, 3, 5, 2, 4)).reshape(4, 4, 128, 32)
    assert np.abs(U3 - np.einsum('ia,ocab,jb->ijoc', G, w3.astype(np.float64), G)).max() < 1e-6
    t5 = ops.winograd_tile_weights(w3, blocks=5)
    U5 = np.transpose(t5.reshape(4, 5, 8, 2, 4, 16, 4), (1, 6, 0, 3, 5, 2, 4)).reshape(5, 4, 128, 32)
    assert np.array_equal(U5[:4], U3) and np.abs(U5[4] - np.einsum('ocb,jb->joc', w3[:, :, 1].astype(np.float64), G)).max() < 1e-6
    got5 = np.empty(20 * 128 * 32, np.float32)
    assert _lib.lib().buf_winograd_tile_filters(w3.ctypes.data_as(C.c_void_p), 128, 32, 2, 5, got5.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(got5, t5)
    got3 = np.empty(16 * 128 * 32, np.float32)
    assert _lib.lib().buf_winograd_tile_weights(w3.ctypes.data_as(C.c_void_p), 128, 32, got3.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(got3, t3)


def test_split_filter_tiling_is_the_f16_pair_of_every_weight():
    """buf_split_tile_filters (host side of csrc/convnet_h3.hip): un-tiling gives hi = f16(w) and lo' = f16((w - hi) 2^11) bit for
    bit as numpy rounds them (round to nearest even, subnormals kept), hi + 2^-11 lo' reproduces w to 2^-23 |w| (+ the f16
    subnormal floor 2^-36), channels beyond Cin are zero, and weights outside the f16 range are rejected."""
    from buffer_amd import _lib, ops
    rng = np.random.default_rng(5)
    for cout, cin in ((64, 48), (32, 32), (128, 128)):
        w = (rng.standard_normal((cout, cin, 3, 3)) * rng.choice([1e-7, 1e-4, 0.05, 3.0], size=(cout, cin, 3, 3))).astype(np.float32)
        w[0, 0, 0, 0], w[1, 0, 0, 0], w[2, 0, 0, 0] = 0.0, 6.0e-8, 65503.0                         # zero, f16-subnormal hi, near the top
        KS = (cin + 31) // 32
        t = ops.split_tile_filters(w).reshape(cout // 32, 9, KS, 2, 2, 4, 16, 8)                 # [g, tap, ks, n2, plane, kg, row, i]
        t = np.transpose(t, (4, 0, 3, 6, 2, 5, 7, 1)).reshape(2, cout, KS * 32, 9)              # [plane, o, c, tap]
        hi = w.reshape(cout, cin, 9).astype(np.float16)
        lo = ((w.reshape(cout, cin, 9) - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
        assert np.array_equal(t[0, :, :cin], hi.view(np.uint16)) and np.array_equal(t[1, :, :cin], lo.view(np.uint16))
        assert not t[:, :, cin:].any()
        back = hi.astype(np.float64) + lo.astype(np.float64) / 2048
        assert np.all(np.abs(back - w.reshape(cout, cin, 9)) <= np.abs(w.reshape(cout, cin, 9)) * 2.0 ** -23 + 2.0 ** -36)
    w[3, 1, 1, 1] = 7.0e4
    with pytest.raises(_lib.BufferHipError):
        ops.split_tile_filters(w)


def test_cnn_entry_points_reject_unsupported_stacks_before_any_device_work():
    """argument checks of the two descriptor-CNN entry points (no GPU needed: both return before the first HIP call): a wide layer
    behind a 32-output layer (the fp32 Winograd kernel's K-split hand-over overwrites the padding zeros of channels
    64..95), Cin 65..96 in the split kernel (three k-steps: not built), a last layer that is not 32 wide."""
    import ctypes as C
    from buffer_amd import _lib
    L = _lib.lib()
    dummy = (C.c_float * 4)()
    ptrs = (C.c_void_p * 8)(*[C.addressof(dummy)] * 8)
    relu = (C.c_int * 8)(1, 1, 1, 1, 1, 1, 1, 0)
    ints = lambda *v: (C.c_int * 8)(*v)
    x = C.addressof(dummy)
    ok_in, ok_out = ints(48, 64, 64, 128, 128, 64, 64, 32), ints(64, 64, 128, 128, 64, 64, 32, 32)
    assert L.buf_cylindrical_net_wg(x, 0, ptrs, ptrs, ok_in, ok_out, relu, x, None) == 0          # nothing to do
    assert L.buf_cylindrical_net_split(x, 0, ptrs, ptrs, ok_in, ok_out, relu, x, None, None) == 0
    rc = L.buf_cylindrical_net_wg(x, 2, ptrs, ptrs, ints(48, 64, 32, 64, 128, 128, 64, 32), ints(64, 32, 64, 128, 128, 64, 32, 32), relu, x, None)
    assert rc == -1 and b"may follow a 32-output layer" in L.buf_last_error()
    rc = L.buf_cylindrical_net_split(x, 2, ptrs, ptrs, ints(80, 64, 64, 128, 128, 64, 64, 32), ok_out, relu, x, None, None)
    assert rc == -1 and b"Cin 65..96" in L.buf_last_error()
    rc = L.buf_cylindrical_net_split(x, 2, ptrs, ptrs, ok_in, ints(64, 64, 128, 128, 64, 64, 32, 64), relu, x, None, None)
    assert rc == -1
    rc = L.buf_cylindrical_net_split(x, 2, ptrs, ptrs, ints(48, 64, 64, 128, 128, 64, 64, 64), ok_out, relu, x, None, None)
    assert rc == -1 and b"width mismatch" in L.buf_last_error()


def test_split_gemm_tiling_general_form():
    """buf_split_tile_gemm (host side of csrc/costnet_h3.hip): groups of nt 16-output tiles, any tap count, Cout padded to whole
    groups (layer 9 of the cost net: 20 -> 32), Cin to whole k-steps; with nt = 2 and 9 taps it is buf_split_tile_filters."""
    from buffer_amd import ops
    rng = np.random.default_rng(9)
    for cout, cin, ntaps, nt in ((20, 32, 4, 1), (64, 96, 9, 2), (32, 32, 15, 2), (32, 64, 9, 1)):
        w = (rng.standard_normal((cout, cin, ntaps)) * 0.1).astype(np.float32)
        cpad, KS = -(-cout // (16 * nt)) * 16 * nt, (cin + 31) // 32
        t = ops.split_tile_gemm(w, nt).reshape(cpad // (16 * nt), ntaps, KS, nt, 2, 4, 16, 8)       # [g, tap, ks, n2, plane, kg, row, i]
        t = np.transpose(t, (4, 0, 3, 6, 2, 5, 7, 1)).reshape(2, cpad, KS * 32, ntaps)             # [plane, o, c, tap]
        hi = w.astype(np.float16)
        lo = ((w - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
        assert np.array_equal(t[0, :cout, :cin], hi.view(np.uint16)) and np.array_equal(t[1, :cout, :cin], lo.view(np.uint16))
        assert not t[:, cout:].any() and not t[:, :, cin:].any()
    w = (rng.standard_normal((64, 48, 3, 3)) * 0.1).astype(np.float32)
    assert np.array_equal(ops.split_tile_filters(w), ops.split_tile_gemm(w.reshape(64, 48, 9), 2))


def test_traffic_summary_keeps_full_step_launches_only(tmp_path):
    """tools/make_traffic.py on a synthetic counter file: a 64-patch probe launch of k_cyl_net_wg beside five full-step launches
    (round 4 averaged it in and under-reported the kernel's traffic by 1/6) must be dropped, and counted as dropped."""
    import csv
    hdr = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
           "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
           "Start_Timestamp", "End_Timestamp"]
    patches, pairs = 2 * 100 * 4, 4

    def write(path, counter, per_patch_kb):
        with open(path, 'w', newline='') as f:
            w = csv.writer(f)
            w.writerow(hdr)
            d = 0
            for k in range(5):                                  # full-step launches: grid = patches x 256 threads
                d += 1
                w.writerow([d, d, "Agent 2", 1, 1, 1, patches * 256, 8, "k_cyl_net_wg(float const*, CylWgParams, float*)", 256, 0, 0, 128, 128, 32,
                            counter, per_patch_kb * patches, 0, 1])
            d += 1                                              # the 64-patch probe launch
            w.writerow([d, d, "Agent 2", 1, 1, 1, 64 * 256, 8, "k_cyl_net_wg(float const*, CylWgParams, float*)", 256, 0, 0, 128, 128, 32,
                        counter, per_patch_kb * 64, 0, 1])
            for k in range(3):                                  # a kernel whose launches legitimately differ in size: all kept
                d += 1
                w.writerow([d, d, "Agent 2", 1, 1, 1, (k + 1) * 1024, 9, "k_cost_net(float const*, float const*, CostNetParams, float*)", 256, 0, 0, 128,
                            128, 32, counter, 10.0 * (k + 1), 0, 1])
    write(tmp_path / 'fetch.csv', 'FETCH_SIZE', 14.0)          # KB per patch (the raw counter; doubled by the tool)
    write(tmp_path / 'write.csv', 'WRITE_SIZE', 17.5)
    bench = {'value': 1.0, 'config': {'pairs_per_step_per_gpu': pairs, 'keypoints_per_fragment': 100},
             'roofline_other': [{'kernel': 'k_cost_net (A13)', 'avg_algorithmic_flops': 51905536.0 * 100}]}
    (tmp_path / 'bench.json').write_text(__import__('json').dumps(bench))
    out = tmp_path / 'traffic.json'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_traffic.py'), str(tmp_path / 'fetch.csv'), str(tmp_path / 'write.csv'),
                        str(tmp_path / 'bench.json'), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    t = __import__('json').load(open(out))['kernels']
    k = t['k_cyl_net_wg']
    assert k['launches_profiled'] == 5 and k['launches_dropped'] == 1
    assert abs(k['hbm_bytes_per_unit'] - (2 * 14.0 + 17.5) * 1024) < 1e-6          # not diluted by the probe launch
    assert t['k_cost_net']['launches_profiled'] == 3 and t['k_cost_net']['launches_dropped'] == 0
