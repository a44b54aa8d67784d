"""Reference-named operator packages (the drop-in boundary).

    import buffer_amd.shims; buffer_amd.shims.install()

puts this directory in front of sys.path so that the imports the reference performs --
  import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling    (ThreeDMatch/dataloader.py:5)
  import cpp_wrappers.cpp_neighbors.radius_neighbors as cpp_neighbors        (ThreeDMatch/dataloader.py:6)
  import pointnet2_ops.pointnet2_utils as pnt2                               (models/BUFFER.py:6)
  from knn_cuda import KNN                                                   (models/BUFFER.py:7)
  from torch_batch_svd import svd                                            (utils/common.py:10)
-- resolve to the gfx950 kernels behind libbuffer_hip.so.  Every shim needs a HIP device; none has
a CPU path.

The reference's model files also import packages that are not operators of the path but must exist for
`models/BUFFER.py`, `models/patch_embedder.py`, `utils/common.py`, `*/config.py` and `*/dataset.py` to import unchanged:
  import kornia.geometry.conversions as Convert          (models/BUFFER.py:10, models/patch_embedder.py:10)
  import open3d as o3d                                    (models/BUFFER.py:11, utils/common.py:1, */dataset.py)
  from easydict import EasyDict as edict                 (*/config.py:2, */dataloader.py:10)
  import nibabel.quaternions as nq                        (ThreeDMatch/test.py:10)
`standins/` holds small replacements for exactly the names used there (open3d's RANSAC / ICP / voxel / normal calls forward
to the device kernels).  That directory goes to the END of sys.path: an environment that has the real package keeps it."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("cpp_wrappers", "pointnet2_ops", "knn_cuda", "torch_batch_svd")
STANDINS = ("kornia", "open3d", "easydict", "nibabel")
_STANDIN_DIR = os.path.join(_HERE, "standins")


def install(standins=True):
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    if standins and _STANDIN_DIR not in sys.path:
        sys.path.append(_STANDIN_DIR)                     # behind site-packages: real kornia / open3d / easydict win
    for n in NAMES:
        m = sys.modules.get(n)
        if m is not None and not getattr(m, "__file__", "").startswith(_HERE):
            raise ImportError(f"{n} is already imported from {getattr(m, '__file__', '?')}")
    return _HERE
