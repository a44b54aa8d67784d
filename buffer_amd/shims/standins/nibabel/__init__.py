"""Stand-in for `nibabel.quaternions.mat2quat` (ThreeDMatch/test.py:10,106)."""
from . import quaternions  # noqa: F401
