"""KITTI odometry test driver: the data side of KITTI/test.py on top of the device pipeline
(counterpart of KITTI/dataset.py:23-118,196-226 test split and KITTI/test.py:43-88).

    <root>/dataset/sequences/<dd>/velodyne/<tttttt>.bin     scans (float32 x, y, z, reflectance)
    <root>/dataset/poses/<dd>.txt                            camera-0 odometry, 12 numbers per frame
    <root>/icp/<drive>_<t0>_<t1>.npy                         ICP-refined ground truth (optional cache)

The reference refines the odometry ground truth with open3d ICP on first use and caches it under icp/
(dataset.py:95-117); here the cached file is used when present, otherwise the same ICP refinement runs on the device
(buffer_amd/icp.py) and fills the cache; --allow-odometry-gt skips the refinement.  The summary reports how many pairs used
which source (`gt_source`).  Host code is file IO and bookkeeping; voxelisation, normals and registration run on the device."""
import glob
import math
import os

import numpy as np
import torch

from . import preprocess
from .threedmatch import items_batched, upload  # the same device-sample -> pipeline-input packing

TEST_DRIVES = (8, 9, 10)                                            # KITTI/test_kitti.txt
VELO2CAM = np.vstack((np.hstack([np.array([7.533745e-03, -9.999714e-01, -6.166020e-04, 1.480249e-02, 7.280733e-04, -9.998902e-01,
                                           9.998621e-01, 7.523790e-03, 1.480755e-02]).reshape(3, 3),
                                 np.array([-4.069766e-03, -7.631618e-02, -2.717806e-01]).reshape(3, 1)]), [0, 0, 0, 1])).T   # dataset.py:203-213


def odometry_to_positions(odometry):
    """dataset.py:216-219"""
    return np.vstack((odometry.reshape(3, 4), [0, 0, 0, 1]))


def select_pairs(scan_ids, positions, min_dist=10.0, window=100):
    """Test pairs of one drive (the rule of KITTI/dataset.py:52-67): starting from the first scan t, the partner is
    the LAST frame before the vehicle has moved more than `min_dist` metres from t (searched over the next `window`
    frames of the pose table); the following pair starts right after the partner.  A frame with no such partner in
    the window is skipped.  positions f64[n_frames,3] (camera-0 translations), scan_ids = frames that have a scan.
    Distances are taken against frame t only (O(window) per step, not the full n x n table)."""
    have = set(int(i) for i in scan_ids)
    pairs, t = [], int(min(have))
    while t in have:
        d = np.sqrt(((positions[t:t + window] - positions[t]) ** 2).sum(-1))
        far = np.flatnonzero(d > min_dist)
        partner = t + int(far[0]) - 1 if far.size else None
        if partner is not None and partner in have:
            pairs.append((t, partner))
            t = partner + 1
        else:
            t += 1          # no partner in the window (or its scan file is missing): move on
    return pairs


class KittiTestSet:
    """KITTIDataset(split='test') (dataset.py:44-70): scan pairs about 10 m apart along the trajectory."""

    def __init__(self, root, drives=TEST_DRIVES, downsample=0.05, voxel_size_0=0.30, max_num_pts=40000,
                 allow_odometry_gt=False):
        self.pc_path = os.path.join(root, 'dataset')
        self.icp_path = os.path.join(root, 'icp')
        self.downsample, self.voxel_size_0, self.max_num_pts = downsample, voxel_size_0, max_num_pts
        self.allow_odometry_gt = allow_odometry_gt
        self.gt_source = {}                                          # pair index -> 'icp-cache' | 'odometry'
        self.files, self._odo = [], {}
        for drive in drives:
            fnames = glob.glob(os.path.join(self.pc_path, 'sequences', '%02d' % drive, 'velodyne', '*.bin'))
            if not fnames:
                raise FileNotFoundError(f'no velodyne scans for drive {drive} under {self.pc_path}')
            scan_ids = [int(os.path.basename(f)[:-4]) for f in fnames]
            positions = np.array([odometry_to_positions(o)[:3, 3] for o in self.odometry(drive)])
            self.files += [(drive, t0, t1) for t0, t1 in select_pairs(scan_ids, positions)]
        if (8, 15, 58) in self.files:                                # "pair (8, 15, 58) is wrong" (dataset.py:69-71)
            self.files.remove((8, 15, 58))

    def odometry(self, drive):
        if drive not in self._odo:
            self._odo[drive] = np.genfromtxt(os.path.join(self.pc_path, 'poses', '%02d.txt' % drive)).reshape(-1, 12)
        return self._odo[drive]

    def __len__(self):
        return len(self.files)

    def scan(self, drive, t):
        fn = os.path.join(self.pc_path, 'sequences', '%02d' % drive, 'velodyne', '%06d.bin' % t)
        return np.ascontiguousarray(np.fromfile(fn, dtype=np.float32).reshape(-1, 4)[:, :3])

    def ground_truth(self, index, device=None):
        """dataset.py:95-117: transform scan t0 -> scan t1.  The reference refines the odometry transform M with
        point-to-point ICP on the raw scans (threshold 0.20 m, <= 200 iterations), stores `M @ T_icp` under
        icp/<drive>_<t0>_<t1>.npy and evaluates against that.  Here: the cached file if present; otherwise the same
        refinement on the device (buffer_amd/icp.py, needs `device`), written to the same cache; the raw odometry
        transform only with `allow_odometry_gt`.  Every pair's source is recorded in `gt_source`."""
        drive, t0, t1 = self.files[index]
        cached = os.path.join(self.icp_path, '%d_%d_%d.npy' % (drive, t0, t1))
        if os.path.exists(cached):
            if self.gt_source.get(index) != 'icp-device':             # (a file this object refined itself keeps its label)
                self.gt_source[index] = 'icp-cache'
            return np.load(cached)
        p0, p1 = (odometry_to_positions(o) for o in self.odometry(drive)[[t0, t1]])
        M = (VELO2CAM @ p0.T @ np.linalg.inv(p1.T) @ np.linalg.inv(VELO2CAM)).T
        if self.allow_odometry_gt:
            self.gt_source[index] = 'odometry'
            return M
        if device is None:
            raise FileNotFoundError(f'{cached} missing: the reference evaluates against ICP-refined poses; call with a device '
                                    f'to refine here, or pass allow_odometry_gt=True (--allow-odometry-gt) for raw odometry')
        from . import icp
        xyz0 = self.scan(drive, t0).astype(np.float64) @ M[:3, :3].T + M[:3, 3]
        T_icp, _, _, _ = icp.icp_point_to_point(torch.from_numpy(xyz0.astype(np.float32)).to(device),
                                                torch.from_numpy(self.scan(drive, t1)).to(device), 0.20, np.eye(4), 200)
        M2 = M @ T_icp                                                 # the reference's composition order (dataset.py:110)
        os.makedirs(self.icp_path, exist_ok=True)
        np.save(cached, M2)
        self.gt_source[index] = 'icp-device'
        return M2

    def raw_pair(self, index):
        drive, t0, t1 = self.files[index]
        return self.scan(drive, t0), self.scan(drive, t1)

    def meta(self, index, device=None):
        drive, t0, t1 = self.files[index]
        return {'src_id': f'{drive:02d}/{t0:06d}', 'tgt_id': f'{drive:02d}/{t1:06d}', 'relt_pose': self.ground_truth(index, device)}

    def item(self, index, device, seed=None):
        """dataset.py:72-178 (test branch) -> sample dict of device tensors (+ relt_pose)."""
        drive, t0, t1 = self.files[index]
        out = self.meta(index, device)
        for side, t in (('src', t0), ('tgt', t1)):
            it = preprocess.prepare_fragment(torch.from_numpy(self.scan(drive, t)).to(device), self.downsample,
                                             self.voxel_size_0, self.max_num_pts, seed=2 * index + (side == 'tgt') if seed is None else seed)
            out[f'{side}_fds_pts'], out[f'{side}_sds_pts'] = it['fds_pts'], it['sds_pts']
        return out


def register_pairs(pipe, dataset, indices, batch=4):
    dev = pipe.device
    poses = []
    idx = list(indices)
    # batches software-pipelined over two HIP streams: reading and pre-processing the fragments of batch i+1 and its keypoint
    # stage run beside the CNN kernels of batch i (BufferPipeline.register_batches; results equal batch-by-batch calls)
    chunks = [idx[lo:lo + batch] for lo in range(0, len(idx), batch)]
    makers = [(lambda ch=ch: [upload(s) for s in items_batched(dataset, ch, dev)]) for ch in chunks]
    for ps in pipe.register_batches(makers, seeds=chunks):
        poses += ps
    return torch.stack(poses) if poses else torch.zeros((0, 4, 4), dtype=torch.float32, device=dev)


def summarize(dataset, poses, rte_thresh=0.3, rre_thresh=1.0):
    """KITTI/test.py:66-88 (note: 0.3 m / 1 degree in the reference's script)."""
    st = []
    for i in range(len(dataset)):
        T, gt = np.asarray(poses[i], np.float64), dataset.ground_truth(i)
        rte = np.linalg.norm(T[:3, 3] - gt[:3, 3])
        rre = np.arccos(np.clip((np.trace(T[:3, :3].T @ gt[:3, :3]) - 1) / 2, -1 + 1e-16, 1 - 1e-16)) * 180 / math.pi
        st.append([rte < rte_thresh and rre < rre_thresh, rte, rre])
    st = np.array(st, np.float64).reshape(-1, 3)
    good = st[:, 0] == 1
    src = list(dataset.gt_source.values())
    return dict(pairs=int(st.shape[0]), recall=float(good.mean()) if st.size else 0.0,
                te=float(st[good, 1].mean()) if good.any() else float('nan'), re=float(st[good, 2].mean()) if good.any() else float('nan'),
                gt_source={k: src.count(k) for k in ('icp-cache', 'icp-device', 'odometry')})


def main(argv=None):
    """python -m buffer_amd.kitti --root <data root>   (one process per GPU under torchrun)"""
    import argparse
    import json
    import time

    import torch.distributed as dist

    from . import dist as bdist
    from .config import KITTI
    from .pipeline import BufferPipeline
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument('--root', required=True)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--limits', default=None)
    ap.add_argument('--allow-odometry-gt', action='store_true',
                    help='evaluate against raw odometry instead of refining it by ICP where <root>/icp/<drive>_<t0>_<t1>.npy is missing')
    a = ap.parse_args(argv)
    rank, world, dev, cdev = bdist.init(int(os.environ.get('LOCAL_RANK', 0)))
    ds = KittiTestSet(a.root, allow_odometry_gt=a.allow_odometry_gt)
    pipe = BufferPipeline(KITTI, dev)
    if a.limits:
        pipe.limits = [int(x) for x in a.limits.split(',')]
    else:
        if rank == 0:
            host = []
            for i in range(min(len(ds), 4)):
                s = ds.item(i, dev)
                host.append({k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in s.items()})
            pipe.calibrate(host)
        pipe.limits = bdist.broadcast_limits(pipe.limits if rank == 0 else [0, 0, 0], device=cdev)
    ids = bdist.shard_indices(len(ds), rank, world)
    t0 = time.perf_counter()
    poses = bdist.gather_poses(ids, register_pairs(pipe, ds, ids, a.batch), len(ds), device=cdev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        out = summarize(ds, poses.cpu().numpy())
        out.update(pairs_per_sec=len(ds) / dt, n_gpus=world, limits=pipe.limits)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
