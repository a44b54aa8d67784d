"""Experiment constants (the reference's easydict configs, */config.py) as frozen dataclasses.
Values from ThreeDMatch/config.py:9-66 and KITTI/config.py."""
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class Config:
    dataset: str = '3DMatch'
    downsample: float = 0.02          # data.downsample (fds voxel)
    voxel_size_0: float = 0.035       # data.voxel_size_0 (sds voxel)
    voxel_size_1: float = 0.035
    max_num_pts: int = 30000
    conv_radius: float = 2.0          # point.conv_radius
    keypts_th: float = 0.1            # point.keypts_th
    num_keypts: int = 1500            # point.num_keypts
    des_r: float = 0.3                # patch.des_r
    num_points_per_patch: int = 512
    rad_n: int = 3
    azi_n: int = 20
    ele_n: int = 7
    delta: float = 0.8
    voxel_sample: int = 10
    dist_th: float = 0.10             # match.dist_th
    inlier_th: float = 1 / 3          # match.inlier_th
    similar_th: float = 0.8
    confidence: float = 0.999
    iter_n: int = 50000
    pose_refine: bool = True          # test.pose_refine
    refine_threshold: float = 0.10    # models/BUFFER.py:395-398
    weights: str = '3dmatch'
    ransac_hypotheses: int = 4096     # build-specific: deterministic GPU RANSAC width
    cnn_arith: str = 'f32'            # build-specific: 'f32' = fp32-MFMA kernels (default) | 'split' = fp32-equivalent split-f16 kernels (opt-in)

    @property
    def scale(self):                  # test.scale = voxel_size_0 / voxel_size_1
        return self.voxel_size_0 / self.voxel_size_1

    @property
    def hist_n(self):                 # ThreeDMatch/dataloader.py:23
        import math
        return int(math.ceil(4 / 3 * math.pi * self.conv_radius ** 3))


THREEDMATCH = Config()
KITTI = replace(Config(), dataset='KITTI', downsample=0.05, voxel_size_0=0.30, voxel_size_1=0.30, max_num_pts=40000,
                keypts_th=0.5, des_r=3.0, dist_th=0.30, inlier_th=2.0, similar_th=0.9, confidence=1.0,
                pose_refine=False, refine_threshold=1.2, weights='kitti')

# Cross-dataset experiments of the reference (generalization/*/config.py): data constants of the TARGET
# data set, weights of the SOURCE data set, test.scale = voxel_size_0 / voxel_size_1 rescaling the neighbour offsets.
THREEDMATCH_TO_KITTI = replace(KITTI, voxel_size_1=0.03, weights='3dmatch')                         # generalization/ThreeD2KITTI
THREEDMATCH_TO_ETH = replace(Config(), dataset='ETH', downsample=0.05, voxel_size_0=0.15, voxel_size_1=0.03, keypts_th=0.5,
                             des_r=1.0, dist_th=0.20, inlier_th=1.5, similar_th=0.9, confidence=1.0, pose_refine=False,
                             refine_threshold=0.10, weights='3dmatch')                                  # generalization/ThreeD2ETH
KITTI_TO_ETH = replace(THREEDMATCH_TO_ETH, voxel_size_1=0.30, inlier_th=2.0, weights='kitti')           # generalization/KITTI2ETH
KITTI_TO_3DLOMATCH = replace(Config(), dataset='3DLoMatch', voxel_size_1=0.30, keypts_th=0.0, weights='kitti')   # generalization/KITTI2ThreeD
PRESETS = {'3DMatch': THREEDMATCH, 'KITTI': KITTI, '3DMatch->KITTI': THREEDMATCH_TO_KITTI, '3DMatch->ETH': THREEDMATCH_TO_ETH,
           'KITTI->ETH': KITTI_TO_ETH, 'KITTI->3DLoMatch': KITTI_TO_3DLOMATCH}
