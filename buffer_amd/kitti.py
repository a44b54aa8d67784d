"""KITTI odometry test driver: the data side of KITTI/test.py on top of the device pipeline
(counterpart of KITTI/dataset.py:23-118,196-226 test split and KITTI/test.py:43-88).

    <root>/dataset/sequences/<dd>/velodyne/<tttttt>.bin     scans (float32 x, y, z, reflectance)
    <root>/dataset/poses/<dd>.txt                            camera-0 odometry, 12 numbers per frame
    <root>/icp/<drive>_<t0>_<t1>.npy                         ICP-refined ground truth (optional cache)

The reference refines the odometry ground truth with open3d ICP on first use and caches it under icp/
(dataset.py:95-117); open3d is not available here, so a cached file is used when present and the plain odometry
transform otherwise.  Host code is file IO and bookkeeping; voxelisation, normals and registration run on the device."""
import glob
import math
import os

import numpy as np
import torch

from . import preprocess
from .threedmatch import upload  # the same device-sample -> pipeline-input packing

TEST_DRIVES = (8, 9, 10)                                            # KITTI/test_kitti.txt
VELO2CAM = np.vstack((np.hstack([np.array([7.533745e-03, -9.999714e-01, -6.166020e-04, 1.480249e-02, 7.280733e-04, -9.998902e-01,
                                           9.998621e-01, 7.523790e-03, 1.480755e-02]).reshape(3, 3),
                                 np.array([-4.069766e-03, -7.631618e-02, -2.717806e-01]).reshape(3, 1)]), [0, 0, 0, 1])).T   # dataset.py:203-213


def odometry_to_positions(odometry):
    """dataset.py:216-219"""
    return np.vstack((odometry.reshape(3, 4), [0, 0, 0, 1]))


class KittiTestSet:
    """KITTIDataset(split='test') (dataset.py:44-70): consecutive scan pairs at least 10 m apart."""

    def __init__(self, root, drives=TEST_DRIVES, downsample=0.05, voxel_size_0=0.30, max_num_pts=40000):
        self.pc_path = os.path.join(root, 'dataset')
        self.icp_path = os.path.join(root, 'icp')
        self.downsample, self.voxel_size_0, self.max_num_pts = downsample, voxel_size_0, max_num_pts
        self.files, self._odo = [], {}
        for drive in drives:
            fnames = glob.glob(os.path.join(self.pc_path, 'sequences', '%02d' % drive, 'velodyne', '*.bin'))
            if not fnames:
                raise FileNotFoundError(f'no velodyne scans for drive {drive} under {self.pc_path}')
            inames = sorted(int(os.path.split(f)[-1][:-4]) for f in fnames)
            have = set(inames)
            all_pos = np.array([odometry_to_positions(o) for o in self.odometry(drive)])
            Ts = all_pos[:, :3, 3]
            pdist = np.sqrt(((Ts.reshape(1, -1, 3) - Ts.reshape(-1, 1, 3)) ** 2).sum(-1))
            more_than_10 = pdist > 10
            curr = inames[0]
            while curr in have:                                      # dataset.py:58-67, verbatim control flow
                nxt = np.where(more_than_10[curr][curr:curr + 100])[0]
                if len(nxt) == 0:
                    curr += 1
                else:
                    nxt = nxt[0] + curr - 1
                if not isinstance(nxt, np.ndarray) and nxt in have:
                    self.files.append((drive, curr, int(nxt)))
                    curr = int(nxt) + 1
        if (8, 15, 58) in self.files:                                # "pair (8, 15, 58) is wrong" (dataset.py:69-71)
            self.files.remove((8, 15, 58))

    def odometry(self, drive):
        if drive not in self._odo:
            self._odo[drive] = np.genfromtxt(os.path.join(self.pc_path, 'poses', '%02d.txt' % drive)).reshape(-1, 12)
        return self._odo[drive]

    def __len__(self):
        return len(self.files)

    def ground_truth(self, index):
        """dataset.py:95-117: ICP-refined transform scan t0 -> scan t1 (cached) or the odometry one."""
        drive, t0, t1 = self.files[index]
        cached = os.path.join(self.icp_path, '%d_%d_%d.npy' % (drive, t0, t1))
        if os.path.exists(cached):
            return np.load(cached)
        p0, p1 = (odometry_to_positions(o) for o in self.odometry(drive)[[t0, t1]])
        return (VELO2CAM @ p0.T @ np.linalg.inv(p1.T) @ np.linalg.inv(VELO2CAM)).T

    def item(self, index, device, seed=None):
        """dataset.py:72-178 (test branch) -> sample dict of device tensors (+ relt_pose)."""
        drive, t0, t1 = self.files[index]
        out = {'src_id': f'{drive:02d}/{t0:06d}', 'tgt_id': f'{drive:02d}/{t1:06d}', 'relt_pose': self.ground_truth(index)}
        for side, t in (('src', t0), ('tgt', t1)):
            fn = os.path.join(self.pc_path, 'sequences', '%02d' % drive, 'velodyne', '%06d.bin' % t)
            xyz = np.fromfile(fn, dtype=np.float32).reshape(-1, 4)[:, :3]
            it = preprocess.prepare_fragment(torch.from_numpy(np.ascontiguousarray(xyz)).to(device), self.downsample,
                                             self.voxel_size_0, self.max_num_pts, seed=2 * index + (side == 'tgt') if seed is None else seed)
            out[f'{side}_fds_pts'], out[f'{side}_sds_pts'] = it['fds_pts'], it['sds_pts']
        return out


def register_pairs(pipe, dataset, indices, batch=4):
    dev = pipe.device
    poses = []
    idx = list(indices)
    for lo in range(0, len(idx), batch):
        chunk = idx[lo:lo + batch]
        poses += pipe.register_batch([upload(dataset.item(i, dev)) for i in chunk], seeds=chunk)
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
    return dict(pairs=int(st.shape[0]), recall=float(good.mean()) if st.size else 0.0,
                te=float(st[good, 1].mean()) if good.any() else float('nan'), re=float(st[good, 2].mean()) if good.any() else float('nan'))


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
    a = ap.parse_args(argv)
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('WORLD_SIZE', 1), ('LOCAL_RANK', 0)))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    ds = KittiTestSet(a.root)
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
        pipe.limits = bdist.broadcast_limits(pipe.limits if rank == 0 else [0, 0, 0], device=dev)
    ids = bdist.shard_indices(len(ds), rank, world)
    t0 = time.perf_counter()
    poses = bdist.gather_poses(ids, register_pairs(pipe, ds, ids, a.batch), len(ds), device=dev)
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
