"""Reference-named operator packages (the drop-in boundary).

    import buffer_amd.shims; buffer_amd.shims.install()

puts this directory in front of sys.path so that the imports the reference performs --
  import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling    (ThreeDMatch/dataloader.py:5)
  import cpp_wrappers.cpp_neighbors.radius_neighbors as cpp_neighbors        (ThreeDMatch/dataloader.py:6)
  import pointnet2_ops.pointnet2_utils as pnt2                               (models/BUFFER.py:6)
  from knn_cuda import KNN                                                   (models/BUFFER.py:7)
  from torch_batch_svd import svd                                            (utils/common.py:10)
-- resolve to the gfx950 kernels behind libbuffer_hip.so.  Every shim needs a HIP device; none has
a CPU path."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("cpp_wrappers", "pointnet2_ops", "knn_cuda", "torch_batch_svd")


def install():
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    for n in NAMES:
        m = sys.modules.get(n)
        if m is not None and not getattr(m, "__file__", "").startswith(_HERE):
            raise ImportError(f"{n} is already imported from {getattr(m, '__file__', '?')}")
    return _HERE
