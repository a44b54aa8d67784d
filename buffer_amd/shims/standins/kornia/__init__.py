"""Stand-in for the one kornia function the reference's model files use (models/BUFFER.py:10,299,
models/patch_embedder.py:10,63): kornia.geometry.conversions.angle_axis_to_rotation_matrix."""
from . import geometry  # noqa: F401
