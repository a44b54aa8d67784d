"""Stand-in for the slice of open3d 0.13 the reference's inference path touches, backed by the gfx950 kernels of
libbuffer_hip.so (open3d==0.13.0 is a pip dependency of the reference, README.md:28; absent from this image).

    geometry.PointCloud                 points / normals / colors, voxel_down_sample, estimate_normals,
                                        orient_normals_towards_camera_location, transform, paint_uniform_color
                                        (ThreeDMatch/dataset.py:91-153, KITTI/dataset.py:102-176, utils/common.py:569-578)
    utility.Vector3dVector, Vector2iVector
    io.read_point_cloud                 PLY vertices (utils/tools.py:6-7)
    pipelines.registration              registration_ransac_based_on_correspondence (models/BUFFER.py:314-326) -> buf_ransac_kabsch,
                                        registration_icp (KITTI/dataset.py:104-107) -> device nearest-neighbour ICP

It is installed behind any real open3d (buffer_amd.shims.install appends this directory to sys.path), so an
environment that has the real package keeps using it.  Point data crosses as numpy arrays exactly as with open3d;
every compute call needs a HIP device (there is no CPU path).  Parity with open3d itself is unpinned."""
from . import geometry, io, pipelines, utility  # noqa: F401

__version__ = "0.13.0+buffer_amd.standin"
