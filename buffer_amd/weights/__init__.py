"""Released BUFFER weights converted to flat f32 arrays keyed by the reference's parameter names
(tests/golden/make_golden.py: the four per-stage best.pth files merged with the reference's substring
stage filter, ThreeDMatch/test.py:207-214)."""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_FILES = {"3dmatch": "3dmatch_06132318.npz", "kitti": "kitti_06050001.npz"}
STAGES = ("Ref", "Desc", "Keypt", "Inlier")


def load_weights(name="3dmatch"):
    """-> dict name -> np.float32 array.  Keys keep the reference names ('Ref.encoder_blocks.0...')."""
    with np.load(os.path.join(_HERE, _FILES[name])) as z:
        return {k: z[k] for k in z.files}


def filter_stage(state, stage):
    """The reference loads a checkpoint by substring: every key that CONTAINS the stage name."""
    return {k: v for k, v in state.items() if stage in k}
