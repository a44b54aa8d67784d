"""Import-level drop-in proof (north_star: "models/BUFFER.py, point_learner.py, patch_embedder.py and
patchnet.py run unchanged on top"): with buffer_amd.shims installed, the reference's own modules import UNCHANGED from
/root/reference, the model constructs and takes the released checkpoints, and every call the reference makes into the
replaced packages binds to the shim's signature (positional count and keyword names taken from the reference's AST).

Build-container test: /root/reference does not travel to the GPU box, where this file skips.  Nothing here computes on
a device -- the kernels behind the same shims are exercised by tests/test_shims_gpu.py and tests/test_standins.py."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'models')), reason='reference tree not present')

# reference files whose calls into the replaced packages are checked, and the local names those packages get there
FILES = ['models/BUFFER.py', 'models/patch_embedder.py', 'utils/common.py', 'utils/tools.py', 'ThreeDMatch/dataloader.py',
         'ThreeDMatch/dataset.py', 'ThreeDMatch/config.py', 'ThreeDMatch/test.py', 'KITTI/dataset.py', 'KITTI/dataloader.py',
         'KITTI/config.py', 'models/KPConv/gcn.py']

PROBE = textwrap.dedent('''
    import ast, importlib, inspect, json, os, sys
    sys.path.insert(0, %(root)r)
    import buffer_amd.shims as shims
    shim_dir = shims.install()
    sys.path.insert(0, %(ref)r)
    out = {"imported": {}, "calls": [], "unbound": []}
    for m in ["ThreeDMatch.config", "KITTI.config", "models.patchnet", "models.vn_layers", "models.point_learner",
              "models.patch_embedder", "models.BUFFER", "utils.common", "utils.tools", "ThreeDMatch.dataset",
              "ThreeDMatch.dataloader", "KITTI.dataset", "KITTI.dataloader", "ThreeDMatch.test"]:
        mod = importlib.import_module(m)
        out["imported"][m] = os.path.realpath(mod.__file__)
    replaced = {}
    for n in shims.NAMES + shims.STANDINS:
        replaced[n] = os.path.realpath(importlib.import_module(n).__file__)
    out["replaced"] = replaced

    # the model builds from the reference's own config and takes the released checkpoints through the stage filter
    import torch
    from ThreeDMatch.config import make_cfg
    from models.BUFFER import buffer
    cfg = make_cfg(); cfg.stage = "test"
    model = buffer(cfg)
    loaded = {}
    for stage in cfg.train.all_stage:
        sd = torch.load(f"%(ref)s/ThreeDMatch/snapshot/06132318/{stage}/best.pth", map_location="cpu")
        new = {k: v for k, v in sd.items() if stage in k}
        res = model.load_state_dict(new, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        loaded.update(new)
    from buffer_amd.weights import load_weights
    mine = load_weights("3dmatch")
    keys = [k for k in loaded if "num_batches_tracked" not in k]
    out["weights_equal"] = sorted(keys) == sorted(mine) and all((loaded[k].numpy() == mine[k]).all() for k in keys)
    out["n_params"] = sum(p.numel() for p in model.parameters())

    # every call into a replaced package, as written in the reference's source
    import pointnet2_ops.pointnet2_utils as pnt2, kornia.geometry.conversions as Convert, open3d, knn_cuda, torch_batch_svd
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
    import cpp_wrappers.cpp_neighbors.radius_neighbors as cpp_neighbors
    import easydict, nibabel.quaternions as nq
    roots = {"pnt2": pnt2, "Convert": Convert, "o3d": open3d, "open3d": open3d, "cpp_subsampling": cpp_subsampling,
             "cpp_neighbors": cpp_neighbors, "KNN": knn_cuda.KNN, "svd": torch_batch_svd.svd, "edict": easydict.EasyDict,
             "nq": nq}
    methods = {n: getattr(open3d.geometry.PointCloud, n) for n in
               ("estimate_normals", "orient_normals_towards_camera_location", "paint_uniform_color", "voxel_down_sample")}

    def chain(node):
        parts = []
        while isinstance(node, ast.Attribute):
            parts.append(node.attr); node = node.value
        if isinstance(node, ast.Name):
            return [node.id] + parts[::-1]
        return None

    for rel in %(files)r:
        tree = ast.parse(open(os.path.join(%(ref)r, rel)).read())
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            c = chain(node.func)
            if any(isinstance(a, ast.Starred) for a in node.args) or any(k.arg is None for k in node.keywords):
                continue
            npos, kws = len(node.args), [k.arg for k in node.keywords]
            target, label, extra_self = None, None, 0
            if c and c[0] in roots:
                obj = roots[c[0]]
                try:
                    for a in c[1:]:
                        obj = getattr(obj, a)
                except AttributeError:
                    out["unbound"].append(f"{rel}:{node.lineno} {'.'.join(c)}: no such attribute in the shim")
                    continue
                target, label = obj, ".".join(c)
            elif c and len(c) >= 2 and c[-1] in methods and c[0] not in ("self", "torch", "np", "F", "nn"):
                target, label, extra_self = methods[c[-1]], "PointCloud." + c[-1], 1
            if target is None:
                continue
            try:
                sig = inspect.signature(target)
                sig.bind(*([object()] * (npos + extra_self)), **{k: object() for k in kws})
                out["calls"].append(f"{rel}:{node.lineno} {label}({npos} positional, keywords {kws})")
            except (TypeError, ValueError) as e:
                out["unbound"].append(f"{rel}:{node.lineno} {label}({npos} positional, keywords {kws}): {e}")
    print("RESULT " + json.dumps(out))
''')


@pytest.fixture(scope='module')
def probe():
    code = PROBE % dict(root=ROOT, ref=REF, files=FILES)
    env = dict(os.environ, PYTHONWARNINGS='ignore')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd='/tmp', env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1]
    return json.loads(line[len('RESULT '):])


def test_reference_modules_import_unchanged_over_the_shims(probe):
    for m, f in probe['imported'].items():
        assert f.startswith(REF + '/'), (m, f)                       # the reference's own files, not copies
    shim_root = os.path.realpath(os.path.join(ROOT, 'buffer_amd', 'shims'))
    for n, f in probe['replaced'].items():
        assert f.startswith(shim_root + '/'), f'{n} resolved to {f}, not to the shim'
    assert probe['n_params'] > 900_000                               # 0.924 M parameters
    assert probe['weights_equal'], 'buffer_amd/weights differs from the released checkpoints loaded through the stage filter'


# display helpers of utils/common.py (mesh_sphere, draw_registration_result, ...): visualisation is out of scope
#  and never executes on the inference path
VISUALISATION = ('open3d.geometry.TriangleMesh', 'open3d.geometry.LineSet', 'open3d.visualization')


def test_every_reference_call_site_binds_to_the_shim_signatures(probe):
    unbound = [u for u in probe['unbound'] if not any(v in u for v in VISUALISATION)]
    assert not unbound, '\n'.join(unbound)
    assert len(probe['unbound']) - len(unbound) <= 4
    calls = '\n'.join(probe['calls'])
    # the call sites SURVEY 8(b) lists must have been seen (guards against a resolver that silently finds nothing)
    for needle in ('models/BUFFER.py', 'pnt2.furthest_point_sample', 'pnt2.gather_operation', 'pnt2.ball_query',
                   'pnt2.grouping_operation', 'KNN(', 'Convert.angle_axis_to_rotation_matrix',
                   'o3d.pipelines.registration.registration_ransac_based_on_correspondence',
                   'o3d.pipelines.registration.registration_icp', 'o3d.utility.Vector2iVector',
                   'open3d.utility.Vector3dVector', 'cpp_subsampling.subsample_batch', 'cpp_neighbors.batch_query',
                   'svd(', 'PointCloud.estimate_normals', 'o3d.geometry.PointCloud.voxel_down_sample',
                   'o3d.io.read_point_cloud', 'nq.mat2quat', 'edict('):
        assert needle in calls, f'no call site of {needle} was resolved'
