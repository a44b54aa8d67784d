"""3DMatch / 3DLoMatch test-set driver: the data side of ThreeDMatch/test.py on top of the device pipeline
(counterpart of ThreeDMatch/dataset.py:47-76,80-162 test split, utils/tools.py:6-7,47-62, test.py:199-308).

    <root>/test/3DMatch/fragments/<scene>/cloud_bin_<k>.ply        fragments
    <root>/test/3DMatch/gt_result/<scene>/gt.log, gt.info          pairs, ground-truth poses, information matrices
    <root>/test/3DLoMatch/<scene>/gt.log, gt.info                  (3DLoMatch pairs over the same fragments)

Host code is file IO and bookkeeping only; voxelisation, normals and registration run on the device
(buffer_amd.preprocess, buffer_amd.pipeline)."""
import os

import numpy as np
import torch

from . import evaluate, preprocess

SCENES = ['7-scenes-redkitchen', 'sun3d-home_at-home_at_scan1_2013_jan_1', 'sun3d-home_md-home_md_scan9_2012_sep_30',
          'sun3d-hotel_uc-scan3', 'sun3d-hotel_umd-maryland_hotel1', 'sun3d-hotel_umd-maryland_hotel3',
          'sun3d-mit_76_studyroom-76-1studyroom2', 'sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika']   # dataset.py:49-58

_PLY_TYPES = {'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'short': 'i2', 'int16': 'i2', 'ushort': 'u2',
              'uint16': 'u2', 'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4', 'float': 'f4', 'float32': 'f4',
              'double': 'f8', 'float64': 'f8'}


def read_ply(path):
    """Vertex positions of a .ply point cloud (ascii, binary_little_endian or binary_big_endian) -> f32[n,3].
    (open3d.io.read_point_cloud in utils/tools.py:6-7; only x, y, z are used by the reference.)"""
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError(f'{path}: not a PLY file')
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f'{path}: truncated PLY header')
            tok = line.decode('ascii', 'replace').split()
            if not tok or tok[0] == 'comment' or tok[0] == 'obj_info':
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == 'property':
                if tok[1] == 'list':
                    elements[-1][2].append((tok[4], 'list', tok[2], tok[3]))
                else:
                    elements[-1][2].append((tok[2], tok[1]))
            elif tok[0] == 'end_header':
                break
        if not elements or elements[0][0] != 'vertex':
            raise ValueError(f'{path}: the first PLY element is not "vertex"')
        _, n, props = elements[0]
        if any(p[1] == 'list' for p in props):
            raise ValueError(f'{path}: list property in the vertex element')
        names = [p[0] for p in props]
        if not all(k in names for k in 'xyz'):
            raise ValueError(f'{path}: vertex element has no x/y/z')
        if fmt == 'ascii':
            rows = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
            cols = [rows[:, names.index(k)] for k in 'xyz']
        elif fmt in ('binary_little_endian', 'binary_big_endian'):
            end = '<' if fmt == 'binary_little_endian' else '>'
            dt = np.dtype([(p[0], end + _PLY_TYPES[p[1]]) for p in props])
            rows = np.frombuffer(f.read(dt.itemsize * n), dtype=dt, count=n)
            cols = [rows[k] for k in 'xyz']
        else:
            raise ValueError(f'{path}: unknown PLY format {fmt}')
    return np.stack(cols, axis=1).astype(np.float32)


def write_ply(path, pts):
    """f32[n,3] -> binary_little_endian PLY (tools and tests)."""
    pts = np.ascontiguousarray(pts, dtype='<f4')
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    with open(path, 'wb') as f:
        f.write(b'ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n'
                b'property float z\nend_header\n' % pts.shape[0])
        f.write(pts.tobytes())


def load_gt_log(gtpath):
    """utils/tools.py:47-62: gt.log -> {'i_j': f64[4,4]} in file order."""
    with open(os.path.join(gtpath, 'gt.log')) as f:
        content = f.readlines()
    result = {}
    for i in range(0, len(content) - 4, 5):
        head = content[i].replace("\n", "").split("\t")[0:3]
        trans = np.array([[float(x) for x in content[i + r].replace("\n", "").split("\t")[0:4]] for r in range(1, 5)])
        result[f'{int(head[0])}_{int(head[1])}'] = trans
    return result


class ThreeDMatchTestSet:
    """ThreeDMatchDataset(split='test') (dataset.py:47-76): the list of (src, tgt, gt) of every scene's gt.log."""

    def __init__(self, root, dataset='3DMatch', scenes=None, downsample=0.02, voxel_size_0=0.035, max_num_pts=30000):
        self.root = os.path.join(root, 'test')
        self.dataset = dataset
        self.gt_root = os.path.join(self.root, dataset, 'gt_result') if dataset == '3DMatch' else os.path.join(self.root, dataset)
        self.downsample, self.voxel_size_0, self.max_num_pts = downsample, voxel_size_0, max_num_pts
        self.files, self.poses = [], []
        for scene in (SCENES if scenes is None else scenes):
            gt = load_gt_log(os.path.join(self.gt_root, scene))
            frag = os.path.join('3DMatch', 'fragments', scene)
            for key, pose in gt.items():
                i, j = key.split('_')
                self.files.append((os.path.join(frag, f'cloud_bin_{i}'), os.path.join(frag, f'cloud_bin_{j}')))
                self.poses.append(pose)

    def __len__(self):
        return len(self.files)

    def raw_pair(self, index):
        """the two fragments of pair `index` as read from disk: (f32[n,3], f32[m,3]) numpy"""
        return tuple(read_ply(os.path.join(self.root, fid + '.ply')) for fid in self.files[index])

    def meta(self, index, device=None):
        src_id, tgt_id = self.files[index]
        return {'src_id': src_id, 'tgt_id': tgt_id, 'relt_pose': np.linalg.inv(self.poses[index])}      # dataset.py:122

    def item(self, index, device, seed=None):
        """dataset.py:80-162 (test branch): read both fragments, two voxel levels, shuffles, normals -- on the device.
        -> the sample dict of the reference, holding DEVICE tensors (+ src_id, tgt_id)."""
        src_id, tgt_id = self.files[index]
        out = self.meta(index)
        for side, fid in (('src', src_id), ('tgt', tgt_id)):
            raw = torch.from_numpy(read_ply(os.path.join(self.root, fid + '.ply'))).to(device)
            it = preprocess.prepare_fragment(raw, self.downsample, self.voxel_size_0, self.max_num_pts,
                                             seed=2 * index + (side == 'tgt') if seed is None else seed)
            out[f'{side}_fds_pts'], out[f'{side}_sds_pts'] = it['fds_pts'], it['sds_pts']
        return out


def items_batched(dataset, indices, device):
    """dataset.item(i, device) for several pairs with the normals of all fragments estimated in ONE stacked pass
    (preprocess.prepare_fragments); pair by pair the result is that of item()."""
    idx = list(indices)
    raws, seeds = [], []
    for i in idx:
        for j, raw in enumerate(dataset.raw_pair(i)):
            raws.append(torch.from_numpy(raw).to(device))
            seeds.append(2 * i + j)
    frs = preprocess.prepare_fragments(raws, dataset.downsample, dataset.voxel_size_0, dataset.max_num_pts, seeds)
    out = []
    for k, i in enumerate(idx):
        s = dataset.meta(i, device)
        s.update(src_fds_pts=frs[2 * k]['fds_pts'], src_sds_pts=frs[2 * k]['sds_pts'],
                 tgt_fds_pts=frs[2 * k + 1]['fds_pts'], tgt_sds_pts=frs[2 * k + 1]['sds_pts'])
        out.append(s)
    return out


def upload(sample):
    """sample of ThreeDMatchTestSet.item (device tensors) -> the inputs BufferPipeline.register takes
    (the device-side twin of pyramid.stack_sample)."""
    src, tgt = sample['src_sds_pts'], sample['tgt_sds_pts']
    return dict(points=torch.cat([src[:, :3], tgt[:, :3]]).contiguous(), features=torch.cat([src[:, 3:], tgt[:, 3:]]).contiguous(),
                lengths=np.array([src.shape[0], tgt.shape[0]], np.int32), src_raw=sample['src_fds_pts'], tgt_raw=sample['tgt_fds_pts'])


def register_pairs(pipe, dataset, indices, batch=32):
    """This rank's share of the pairs through the device pipeline -> f32[k,4,4] (device), in the order of `indices`."""
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


def write_logs(dataset, poses, log_root, log_name):
    """test.py:242-270 over all pairs in dataset order: append the inverse pose to <log_root>/<scene>/<log_name>
    and collect the DGR statistics.  poses f32[n,4,4].  -> list of (success, rte, rre)."""
    stats = []
    for i, (src_id, tgt_id) in enumerate(dataset.files):
        T = np.asarray(poses[i], dtype=np.float64)
        scene = src_id.split(os.sep)[-2]
        evaluate.append_log(os.path.join(log_root, scene, log_name), src_id.split('_')[-1], tgt_id.split('_')[-1], T)
        stats.append(evaluate.dgr_success(T, np.linalg.inv(dataset.poses[i])))
    return stats


def summarize(dataset, stats, log_root, log_name):
    """DGR recall / TE / RE (test.py:278-284) and the Registration Recall over the scenes' logs (:287-308)."""
    st = np.array([[float(a), b, c] for a, b, c in stats], np.float64).reshape(-1, 3)
    good = st[:, 0] == 1
    rr, per_scene = evaluate.registration_recall(dataset.gt_root, log_root, log_name)
    return dict(pairs=int(st.shape[0]), dgr_recall=float(good.mean()) if st.size else 0.0,
                te=float(st[good, 1].mean()) if good.any() else float('nan'),
                re=float(st[good, 2].mean()) if good.any() else float('nan'),
                registration_recall=rr, per_scene=[float(x) for x in per_scene])


def main(argv=None):
    """python -m buffer_amd.threedmatch --root <data root> [--dataset 3DLoMatch]   (one process per GPU under torchrun)"""
    import argparse
    import json
    import time

    import torch.distributed as dist

    from . import dist as bdist
    from .pipeline import BufferPipeline
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument('--root', required=True)
    ap.add_argument('--dataset', default='3DMatch', choices=['3DMatch', '3DLoMatch'])
    ap.add_argument('--log-root', default=None)
    ap.add_argument('--log-name', default=time.strftime('%m%d%H%M') + '.log')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--limits', default=None, help='frozen neighbourhood limits "a,b,c" (default: calibrate like dataloader.py:18-51)')
    a = ap.parse_args(argv)
    rank, world, dev, cdev = bdist.init(int(os.environ.get('LOCAL_RANK', 0)))
    ds = ThreeDMatchTestSet(a.root, a.dataset)
    pipe = BufferPipeline(device=dev)
    if a.limits:
        pipe.limits = [int(x) for x in a.limits.split(',')]
    else:
        if rank == 0:                                        # dataloader.py:18-51 on the first pairs
            host = []
            for i in range(min(len(ds), 8)):
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
        log_root = a.log_root or f'log_{a.dataset}'
        stats = write_logs(ds, poses.cpu().numpy(), log_root, a.log_name)
        out = summarize(ds, stats, log_root, a.log_name)
        out.update(pairs_per_sec=len(ds) / dt, n_gpus=world, limits=pipe.limits)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
