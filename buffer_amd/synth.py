"""Synthetic fragment pairs of 3DMatch / KITTI shape.

The datasets are not available offline, so the benchmark and the parity tests
run on seeded synthetic scenes that reproduce the *shape* of what
ThreeDMatch/dataset.py:80-162 hands to the collate function: a 2.5 cm
("fds", patch support) and a 3.5 cm ("sds", + unit normals oriented to the
sensor at the origin) voxel cloud per fragment, randomly permuted, plus the
ground-truth pose src -> tgt.

Host-side numpy only: this is input generation, not part of the hot path.
"""
import numpy as np


def voxel_down_sample(pts, voxel):
    """Voxel-centroid downsample (stand-in for open3d's voxel_down_sample)."""
    if pts.shape[0] == 0:
        return pts
    mn = pts.min(0)
    key = np.floor((pts - mn) / voxel).astype(np.int64)
    dims = key.max(0) + 1
    flat = key[:, 0] + dims[0] * (key[:, 1] + dims[1] * key[:, 2])
    uniq, inv, cnt = np.unique(flat, return_inverse=True, return_counts=True)
    out = np.zeros((uniq.shape[0], pts.shape[1]), np.float64)
    np.add.at(out, inv, pts)
    return out / cnt[:, None]


def random_rotation(rng, max_angle=np.pi):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _rect(origin, eu, ev, normal):
    return dict(o=np.asarray(origin, float), u=np.asarray(eu, float), v=np.asarray(ev, float),
                n=np.asarray(normal, float))


def make_scene(rng, size=(3.0, 3.0, 2.5), n_boxes=6):
    """Room corner (floor + two walls) plus a few boxes: a list of rectangles."""
    sx, sy, sz = size
    rects = [
        _rect([0, 0, 0], [sx, 0, 0], [0, sy, 0], [0, 0, 1]),      # floor
        _rect([0, 0, 0], [sx, 0, 0], [0, 0, sz], [0, 1, 0]),      # wall y=0
        _rect([0, 0, 0], [0, sy, 0], [0, 0, sz], [1, 0, 0]),      # wall x=0
    ]
    for _ in range(n_boxes):
        w, d, h = rng.uniform(0.25, 0.9, 3)
        x0 = rng.uniform(0.05, sx - w - 0.05)
        y0 = rng.uniform(0.05, sy - d - 0.05)
        z0 = 0.0 if rng.random() < 0.7 else rng.uniform(0.3, 1.2)
        rects += [
            _rect([x0, y0, z0 + h], [w, 0, 0], [0, d, 0], [0, 0, 1]),
            _rect([x0, y0, z0], [w, 0, 0], [0, 0, h], [0, -1, 0]),
            _rect([x0, y0 + d, z0], [w, 0, 0], [0, 0, h], [0, 1, 0]),
            _rect([x0, y0, z0], [0, d, 0], [0, 0, h], [-1, 0, 0]),
            _rect([x0 + w, y0, z0], [0, d, 0], [0, 0, h], [1, 0, 0]),
        ]
    return rects


def sample_scene(rng, rects, n, jitter=0.003):
    area = np.array([np.linalg.norm(np.cross(r['u'], r['v'])) for r in rects])
    which = rng.choice(len(rects), size=n, p=area / area.sum())
    a, b = rng.random(n), rng.random(n)
    o = np.stack([rects[i]['o'] for i in which])
    u = np.stack([rects[i]['u'] for i in which])
    v = np.stack([rects[i]['v'] for i in which])
    nrm = np.stack([rects[i]['n'] for i in which])
    pts = o + a[:, None] * u + b[:, None] * v + rng.normal(scale=jitter, size=(n, 3))
    return pts, nrm


def _with_normals(rng, pts, nrm, sensor):
    """Unit normals oriented towards the sensor (+ small jitter), like
    estimate_normals + orient_normals_towards_camera_location (dataset.py:142-153)."""
    nrm = nrm + rng.normal(scale=0.02, size=nrm.shape)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    flip = np.sum(nrm * (sensor[None] - pts), axis=1) < 0
    nrm[flip] *= -1
    return nrm


def make_pair(seed, n_raw=250_000, fds_voxel=0.025, sds_voxel=0.035, overlap=0.7, size=(1.9, 1.9, 1.7),
              n_boxes=5, max_num_pts=30000):
    """One fragment pair in the sample-dict format of ThreeDMatch/dataset.py:155-161.

    Returns dict(src_fds_pts f64[Nf,3], tgt_fds_pts, src_sds_pts f64[n,6], tgt_sds_pts, relt_pose f64[4,4]).
    Both fragments are expressed in their own sensor frame (sensor at the origin).
    """
    rng = np.random.default_rng(seed)
    rects = make_scene(rng, size, n_boxes)
    out = {}
    sx = size[0]
    # the two fragments see overlapping slabs of the room along x
    width = sx / (2 - overlap)
    windows = [(0.0, width), (sx - width, sx)]
    sensors = [np.array([0.55 * width, 0.6 * size[1], 0.5 * size[2]]),
               np.array([sx - 0.55 * width, 0.55 * size[1], 0.55 * size[2]])]
    poses = []
    for name, (lo, hi), sensor in zip(('src', 'tgt'), windows, sensors):
        pts, nrm = sample_scene(rng, rects, n_raw)
        keep = (pts[:, 0] >= lo) & (pts[:, 0] <= hi)
        pts, nrm = pts[keep], nrm[keep]
        # world -> sensor frame
        R = random_rotation(rng, np.pi if name == 'tgt' else 0.3)
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ sensor
        poses.append(T)
        both = np.concatenate([pts, nrm], 1)
        fds = voxel_down_sample(both, fds_voxel)
        sds = voxel_down_sample(fds, sds_voxel)
        fds_p = fds[:, :3] @ R.T + T[:3, 3]
        sds_p = sds[:, :3] @ R.T + T[:3, 3]
        sds_n = _with_normals(rng, sds_p, sds[:, 3:] @ R.T, np.zeros(3))
        rng.shuffle(fds_p)
        perm = rng.permutation(sds_p.shape[0])
        sds_p, sds_n = sds_p[perm], sds_n[perm]
        if sds_p.shape[0] > max_num_pts:
            sds_p, sds_n = sds_p[:max_num_pts], sds_n[:max_num_pts]
        out[f'{name}_fds_pts'] = fds_p
        out[f'{name}_sds_pts'] = np.concatenate([sds_p, sds_n], 1)
    out['relt_pose'] = poses[1] @ np.linalg.inv(poses[0])
    out['src_id'] = f'synth/{seed}_0'
    out['tgt_id'] = f'synth/{seed}_1'
    return out


def make_config1_pair(seed=7, n=5000):
    """BASELINE config #1: two clouds of exactly n points on three
    mutually orthogonal faces, sigma = 3 mm jitter, all coordinates > 0.4 m;
    tgt = rigid transform of an independent sample (||t|| <= 1 m)."""
    rng = np.random.default_rng(seed)
    rects = make_scene(rng, (2.0, 2.0, 2.0), n_boxes=0)

    def cloud():
        p, nrm = sample_scene(rng, rects, n)
        return p + 0.45, nrm

    src, src_n = cloud()
    tgt, tgt_n = cloud()
    R = random_rotation(rng, 0.5)
    t = rng.normal(size=3)
    t *= rng.uniform(0, 1) / np.linalg.norm(t)
    c = tgt.mean(0)
    tgt = (tgt - c) @ R.T + c + t
    tgt_n = tgt_n @ R.T
    tgt = tgt - np.minimum(tgt.min(0) - 0.45, 0)
    pose = np.eye(4)
    pose[:3, :3] = R
    pose[:3, 3] = c + t - R @ c
    return dict(src_fds_pts=src.copy(), tgt_fds_pts=tgt.copy(),
                src_sds_pts=np.concatenate([src, _with_normals(rng, src, src_n, np.zeros(3))], 1),
                tgt_sds_pts=np.concatenate([tgt, _with_normals(rng, tgt, tgt_n, np.zeros(3))], 1),
                relt_pose=pose, src_id='config1/src', tgt_id='config1/tgt')


def make_kitti_pair(seed, beams=64, az_steps=1900, fds_voxel=0.05, sds_voxel=0.30):
    """Ring-pattern LiDAR scans (~120k returns) over a ground plane with boxes."""
    rng = np.random.default_rng(seed)
    boxes = [(rng.uniform(-60, 60), rng.uniform(-60, 60), rng.uniform(2, 10), rng.uniform(2, 10),
              rng.uniform(1.5, 6)) for _ in range(40)]
    out = {}
    poses = []
    for name, pos in (('src', np.zeros(3)), ('tgt', np.array([rng.uniform(5, 10), rng.uniform(-1, 1), 0.0]))):
        el = np.deg2rad(np.linspace(-24.8, 2.0, beams))
        az = np.linspace(0, 2 * np.pi, az_steps, endpoint=False)
        E, A = np.meshgrid(el, az, indexing='ij')
        d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
        h = 1.73
        o = pos + np.array([0, 0, h])
        with np.errstate(divide='ignore', invalid='ignore'):
            t = np.where(d[:, 2] < 0, -o[2] / d[:, 2], np.inf)
        for (bx, by, w, l, hh) in boxes:
            lo = np.array([bx - w / 2, by - l / 2, 0.0])
            hi = np.array([bx + w / 2, by + l / 2, hh])
            with np.errstate(divide='ignore', invalid='ignore'):
                t1 = (lo - o) / d
                t2 = (hi - o) / d
            tn = np.nanmax(np.minimum(t1, t2), 1)
            tf = np.nanmin(np.maximum(t1, t2), 1)
            hit = (tn < tf) & (tn > 0)
            t = np.where(hit & (tn < t), tn, t)
        ok = (t > 3) & (t < 80)
        pts = o + d[ok] * t[ok, None] + rng.normal(scale=0.01, size=(ok.sum(), 3))
        yaw = rng.uniform(-0.1, 0.1) if name == 'tgt' else 0.0
        R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ o
        poses.append(T)
        p = pts @ R.T + T[:3, 3]
        fds = voxel_down_sample(p, fds_voxel)
        sds = voxel_down_sample(fds, sds_voxel)
        nrm = np.tile(np.array([[0, 0, 1.0]]), (sds.shape[0], 1))
        rng.shuffle(fds)
        sds = sds[rng.permutation(sds.shape[0])]
        out[f'{name}_fds_pts'] = fds
        out[f'{name}_sds_pts'] = np.concatenate([sds, _with_normals(rng, sds, nrm, np.zeros(3))], 1)
    out['relt_pose'] = poses[1] @ np.linalg.inv(poses[0])
    out['src_id'] = f'kitti_synth/{seed}_0'
    out['tgt_id'] = f'kitti_synth/{seed}_1'
    return out


def make_raw_pair_device(seed, overlap, device, n_raw=250_000, size=(1.9, 1.9, 1.7), n_boxes=5, jitter=0.003):
    """A fragment pair as RAW (un-voxelised) clouds in device memory: what a 3DMatch .ply holds before
    ThreeDMatch/dataset.py:91-153 voxelises it.  Same room / window / sensor construction as make_pair(); the scene
    comes from the seeded numpy generator, the ~n_raw surface samples per fragment from a seeded torch generator on the
    device (a 1623-pair stream is a few seconds of generation instead of minutes).
    -> dict(src_raw f32[n,3], tgt_raw f32[m,3] (device, each in its own sensor frame), relt_pose f64[4,4] numpy src->tgt,
            overlap_pts f32[k,3] (device): tgt-frame points of the shared slab, for the RMSE-proxy information matrix)."""
    import torch
    rng = np.random.default_rng(seed)
    rects = make_scene(rng, size, n_boxes)
    g = torch.Generator(device=device).manual_seed(int(seed))
    o = torch.tensor(np.stack([r['o'] for r in rects]), dtype=torch.float32, device=device)
    u = torch.tensor(np.stack([r['u'] for r in rects]), dtype=torch.float32, device=device)
    v = torch.tensor(np.stack([r['v'] for r in rects]), dtype=torch.float32, device=device)
    area = torch.linalg.norm(torch.linalg.cross(u, v), dim=1)
    sx = size[0]
    width = sx / (2 - overlap)
    windows = [(0.0, width), (sx - width, sx)]
    sensors = [np.array([0.55 * width, 0.6 * size[1], 0.5 * size[2]]), np.array([sx - 0.55 * width, 0.55 * size[1], 0.55 * size[2]])]
    out, poses = {}, []
    for name, (lo, hi), sensor in zip(('src', 'tgt'), windows, sensors):
        which = torch.multinomial(area / area.sum(), n_raw, replacement=True, generator=g)
        ab = torch.rand((n_raw, 2), generator=g, device=device)
        pts = o[which] + ab[:, :1] * u[which] + ab[:, 1:] * v[which] + jitter * torch.randn((n_raw, 3), generator=g, device=device)
        pts = pts[(pts[:, 0] >= lo) & (pts[:, 0] <= hi)]
        R = random_rotation(rng, np.pi if name == 'tgt' else 0.3)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R, -R @ sensor
        poses.append(T)
        Rt = torch.tensor(R, dtype=torch.float32, device=device)
        tt = torch.tensor(T[:3, 3], dtype=torch.float32, device=device)
        if name == 'tgt':
            shared = pts[(pts[:, 0] >= windows[1][0]) & (pts[:, 0] <= windows[0][1])]
            out['overlap_pts'] = (shared[:: max(1, shared.shape[0] // 4096)] @ Rt.T + tt).contiguous()
        out[f'{name}_raw'] = (pts @ Rt.T + tt).contiguous()
    out['relt_pose'] = poses[1] @ np.linalg.inv(poses[0])
    return out


def information_matrix(points):
    """6x6 information matrix of a set of overlap points p (fragment-j frame), in the convention of the 3DMatch
    gt.info files that ThreeDMatch/test.py:92-111 consumes: for er = [t, q_xyz] (translation and quaternion vector part of
    a small residual transform) the displacement of p is t + 2 q x p = [I, -2[p]x] er, so  er^T (sum J^T J) er / info[0,0]
    is the mean squared displacement over the overlap -- the RMSE proxy thresholded at 0.2^2 by the RR metric."""
    p = np.asarray(points, np.float64).reshape(-1, 3)
    J = np.zeros((p.shape[0], 3, 6))
    J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1.0
    J[:, 0, 4], J[:, 0, 5] = 2 * p[:, 2], -2 * p[:, 1]
    J[:, 1, 3], J[:, 1, 5] = -2 * p[:, 2], 2 * p[:, 0]
    J[:, 2, 3], J[:, 2, 4] = 2 * p[:, 1], -2 * p[:, 0]
    return np.einsum('nij,nik->jk', J, J)
