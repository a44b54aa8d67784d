"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (imported from /root/reference, this
container only) on seeded synthetic pairs, with the released 3DMatch weights.

Runs only where /root/reference is mounted; the fixtures (inputs + expected outputs, data only) and
this script are what gets committed.  Third-party packages the reference imports but that are not
installed here are stubbed:
  * cpp_wrappers.*            -> the reference's own C++ cores compiled in place (oracle/_ref)
  * pointnet2_ops, knn_cuda   -> oracle/cpu.py (plain-C restatement of the upstream semantics; these
                                 CUDA packages are not under /root/reference -> parity unpinned there)
  * torch_batch_svd, kornia, easydict, open3d, nibabel, tensorboardX -> minimal stand-ins
  * torch.Tensor.cuda         -> identity

Usage: python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = '/root/reference'
GOLD = os.path.join(ROOT, 'tests', 'golden')

from oracle import cpu  # noqa: E402
from buffer_amd import synth  # noqa: E402


def install_stubs():
    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Sink(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            s = Sink(self.__name__ + '.' + k)
            setattr(self, k, s)
            return s

        def __call__(self, *a, **k):
            return None

    mod('easydict', EasyDict=EasyDict)
    for n in ('open3d', 'tensorboardX'):
        sys.modules[n] = Sink(n)

    def mat2quat(M):
        # nibabel.quaternions.mat2quat [recalled]: Bar-Itzhack eigen method, w >= 0
        Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, float).flat
        K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0], [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                      [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                      [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
        vals, vecs = np.linalg.eigh(K)
        q = vecs[[3, 0, 1, 2], np.argmax(vals)]
        return q * -1 if q[0] < 0 else q

    nqm = mod('nibabel.quaternions', mat2quat=mat2quat)
    mod('nibabel', quaternions=nqm)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        sys.modules['matplotlib'] = Sink('matplotlib')
        sys.modules['matplotlib.pyplot'] = Sink('matplotlib.pyplot')
        sys.modules['matplotlib.cm'] = Sink('matplotlib.cm')

    def angle_axis_to_rotation_matrix(aa):
        # kornia: Rodrigues; BUFFER only feeds (0,0,theta) with theta >= 1e-6
        theta = aa.norm(dim=1, keepdim=True)
        k = aa / theta
        K = torch.zeros(aa.shape[0], 3, 3)
        K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
        K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
        K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
        th = theta.view(-1, 1, 1)
        return torch.eye(3)[None] + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)

    conv = mod('kornia.geometry.conversions', angle_axis_to_rotation_matrix=angle_axis_to_rotation_matrix)
    geo = mod('kornia.geometry', conversions=conv)
    mod('kornia', geometry=geo)
    mod('torch_batch_svd', svd=torch.svd)

    class KNN:
        def __init__(self, k, transpose_mode=False):
            self.k, self.t = k, transpose_mode

        def __call__(self, ref, query):
            assert self.t
            d, i = cpu.knn(ref.detach().numpy(), query.detach().numpy(), self.k)
            return torch.from_numpy(d), torch.from_numpy(i)

    mod('knn_cuda', KNN=KNN)

    pu = mod('pointnet2_ops.pointnet2_utils',
             furthest_point_sample=lambda xyz, m: torch.from_numpy(cpu.fps(xyz.detach().numpy(), m)),
             gather_operation=lambda f, i: torch.from_numpy(cpu.gather_operation(f.detach().numpy(), i.numpy())),
             ball_query=lambda r, n, xyz, new: torch.from_numpy(
                 cpu.ball_query(r, n, xyz.detach().numpy(), new.detach().numpy())),
             grouping_operation=lambda f, i: torch.from_numpy(
                 cpu.grouping_operation(f.detach().numpy(), i.numpy())))
    mod('pointnet2_ops', pointnet2_utils=pu)

    def subsample_batch(points, batches, sampleDl=0.1, max_p=0, verbose=0, **kw):
        return cpu.ref_grid_subsample_batch(np.asarray(points, np.float32), np.asarray(batches, np.int32),
                                            sampleDl, max_p)

    def batch_query(queries, supports, q_batches, s_batches, radius=0.1):
        return cpu.ref_radius_neighbors(np.asarray(queries, np.float32), np.asarray(supports, np.float32),
                                        np.asarray(q_batches, np.int32), np.asarray(s_batches, np.int32), radius)

    gsm = mod('cpp_wrappers.cpp_subsampling.grid_subsampling', subsample_batch=subsample_batch)
    rnm = mod('cpp_wrappers.cpp_neighbors.radius_neighbors', batch_query=batch_query)
    cs = mod('cpp_wrappers.cpp_subsampling', grid_subsampling=gsm)
    cn = mod('cpp_wrappers.cpp_neighbors', radius_neighbors=rnm)
    mod('cpp_wrappers', cpp_subsampling=cs, cpp_neighbors=cn)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def load_reference_model(cfg_module='ThreeDMatch', exp='06132318'):
    import importlib
    cfgm = importlib.import_module(f'{cfg_module}.config')
    cfg = cfgm.make_cfg()
    cfg.stage = 'test'
    from models.BUFFER import buffer
    model = buffer(cfg)
    merged = {}
    for stage in cfg.train.all_stage:                        # ThreeDMatch/test.py:207-214
        sd = torch.load(f'{REF}/{cfg_module}/snapshot/{exp}/{stage}/best.pth', map_location='cpu')
        new = {k: v for k, v in sd.items() if stage in k}
        model.load_state_dict(new, strict=False)
        merged.update(new)
    model.eval()
    return cfg, model, merged


def save_weights(merged, name):
    os.makedirs(os.path.join(ROOT, 'buffer_amd', 'weights'), exist_ok=True)
    arrs = {k: v.numpy() for k, v in merged.items() if 'num_batches_tracked' not in k}
    np.savez_compressed(os.path.join(ROOT, 'buffer_amd', 'weights', name), **arrs)
    return arrs


def main():
    cpu.build(ref=True)
    install_stubs()
    torch.manual_seed(0)
    np.random.seed(0)
    os.makedirs(GOLD, exist_ok=True)
    from ThreeDMatch import dataloader as dl

    cfg, model, merged = load_reference_model()
    save_weights(merged, '3dmatch_06132318.npz')
    _, _, merged_k = load_reference_model('KITTI', '06050001')
    save_weights(merged_k, 'kitti_06050001.npz')
    sys.modules.pop('KITTI.config', None)

    # ---- F1 / F2: pyramid from the compiled reference cores via the reference's own collate -------
    def pyramid_fixture(sample, name):
        limits = dl.calibrate_neighbors([sample], cfg, dl.collate_fn_descriptor)
        batch = dl.collate_fn_descriptor([sample], cfg, limits)
        out = {'limits': np.asarray(limits, np.int32)}
        for k in ('src_fds_pts', 'tgt_fds_pts', 'src_sds_pts', 'tgt_sds_pts', 'relt_pose'):
            out['in_' + k] = sample[k]
        for l in range(3):
            out[f'points_{l}'] = batch['points'][l].numpy()
            out[f'neighbors_{l}'] = batch['neighbors'][l].numpy().astype(np.int32)
            out[f'pools_{l}'] = batch['pools'][l].numpy().astype(np.int32)
            out[f'upsamples_{l}'] = batch['upsamples'][l].numpy().astype(np.int32)
            out[f'lengths_{l}'] = batch['stack_lengths'][l].numpy().astype(np.int32)
        np.savez_compressed(os.path.join(GOLD, name), **out)
        return limits, batch

    tiny = synth.make_pair(11, n_raw=40_000, size=(1.0, 1.0, 0.9), n_boxes=3)
    limits, batch = pyramid_fixture(tiny, 'pyramid_tiny.npz')
    print('tiny: sds', tiny['src_sds_pts'].shape, tiny['tgt_sds_pts'].shape, 'limits', limits,
          'layers', [p.shape[0] for p in batch['points']])
    c1 = synth.make_config1_pair()
    lim1, b1 = pyramid_fixture(c1, 'pyramid_5k.npz')
    print('config1: limits', lim1, 'layers', [p.shape[0] for p in b1['points']],
          'maxK', [int((n < n.shape[0]).sum(1).max()) for n in b1['neighbors']])

    # ---- F3: point-wise learner (Ref + Keypt) on the tiny pair ----------------------------------
    with torch.no_grad():
        acts = []
        hooks = [blk.register_forward_hook(lambda m, i, o: acts.append(o.numpy().copy()))
                 for blk in model.Ref.encoder_blocks]
        axis, eps, branch = model.Ref(batch)
        for h in hooks:
            h.remove()
        bottle = branch['bottle_feature']
        skips = [s.clone() for s in branch['skip_feature']]
        score = model.Keypt(batch, {'bottle_feature': bottle, 'skip_feature': list(branch['skip_feature'])})
    f3 = dict(axis=axis.numpy(), eps=eps.numpy(), score=score.numpy(), bottle=bottle.numpy(),
              skip0=skips[0].numpy(), skip1=skips[1].numpy(), features=batch['features'].numpy())
    for i, a in enumerate(acts):
        f3[f'block{i}'] = a
    np.savez_compressed(os.path.join(GOLD, 'point_learner_tiny.npz'), **f3)
    print('F3 axis', axis.shape, 'score range', float(score.min()), float(score.max()),
          'kept', int((score[:, 0] > cfg.point.keypts_th).sum()))

    # ---- F4: patch embedder on 64 FPS keypoints of the src cloud ---------------------------------
    import torch.nn.functional as F
    n_src = int(batch['stack_lengths'][0][0])
    src_pts = batch['src_pcd']
    src_axis = F.normalize(axis[:n_src], p=2, dim=1)
    mask = (torch.sum(-src_axis * src_pts, dim=1) < 0).float().unsqueeze(1)
    src_axis = src_axis * (1 - mask) - src_axis * mask
    P = 64
    fps_idx = torch.from_numpy(cpu.fps(src_pts[None].numpy(), P)).long()[0]
    kpts, kaxis = src_pts[fps_idx], src_axis[fps_idx]
    raw = batch['src_pcd_raw']
    perm = np.random.RandomState(5).permutation(raw.shape[0])
    orig_choice = np.random.choice
    np.random.choice = lambda n, size=None, replace=True: perm           # pin select_patches' shuffle
    inter = {}
    orig_spt = model.Desc.SPT

    def spy_spt(delta_x, des_r, voxel_r):
        out = orig_spt(delta_x, des_r, voxel_r)
        inter['spt'] = out.numpy().copy()
        return out

    model.Desc.SPT = spy_spt
    with torch.no_grad():
        out = model.Desc(raw[None], kpts[None], kaxis[None])
    np.random.choice = orig_choice
    model.Desc.SPT = orig_spt
    spt_full = inter['spt']
    np.savez_compressed(
        os.path.join(GOLD, 'desc_tiny.npz'), raw=raw.numpy(), kpts=kpts.numpy(), kaxis=kaxis.numpy(),
        perm=perm.astype(np.int64), fps_idx=fps_idx.numpy().astype(np.int32),
        patches=out['patches'].numpy(), R=out['R'].numpy(), rand_axis=out['rand_axis'].numpy(),
        desc=out['desc'].numpy(), equi=out['equi'].numpy(),
        spt_first8=spt_full[:8], spt_sum=spt_full.sum((2, 3)), spt_abs_sum=np.abs(spt_full).sum((2, 3)))
    print('F4 desc', out['desc'].shape, 'equi', out['equi'].shape)

    # ---- F5: matching / inlier / hypotheses / refinement on 64+64 keypoints ------------------------
    tgt_pts = batch['tgt_pcd']
    tgt_axis = F.normalize(axis[n_src:], p=2, dim=1)
    mask = (torch.sum(-tgt_axis * tgt_pts, dim=1) < 0).float().unsqueeze(1)
    tgt_axis = tgt_axis * (1 - mask) - tgt_axis * mask
    # keypoints of tgt = gt-transformed src keypoints snapped to the nearest tgt point (so matches exist)
    gt = batch['relt_pose']
    moved = kpts @ gt[:3, :3].T + gt[:3, 3]
    nn_idx = torch.cdist(moved, tgt_pts).argmin(1)
    nn_idx = torch.unique(nn_idx)
    tk, ta = tgt_pts[nn_idx], tgt_axis[nn_idx]
    traw = batch['tgt_pcd_raw']
    perm_t = np.random.RandomState(6).permutation(traw.shape[0])
    np.random.choice = lambda n, size=None, replace=True: perm_t
    with torch.no_grad():
        tout = model.Desc(traw[None], tk[None], ta[None])
    np.random.choice = orig_choice
    with torch.no_grad():
        s_mids, t_mids = model.mutual_matching(out['desc'], tout['desc'])
        ss_kpts, tt_kpts = kpts[s_mids], tk[t_mids]
        ss_equi, tt_equi = out['equi'][s_mids], tout['equi'][t_mids]
        ss_R, tt_R = out['R'][s_mids], tout['R'][t_mids]
        ind = model.Inlier(ss_equi[:, :, 1:cfg.patch.ele_n - 1], tt_equi[:, :, 1:cfg.patch.ele_n - 1])
        import kornia.geometry.conversions as Convert
        angle = ind * 2 * np.pi / cfg.patch.azi_n + 1e-6                 # models/BUFFER.py:295-311
        angle_axis = torch.zeros_like(ss_kpts)
        angle_axis[:, -1] = 1
        angle_axis = angle_axis * angle[:, None]
        azi_R = Convert.angle_axis_to_rotation_matrix(angle_axis)
        Rh = tt_R @ azi_R @ ss_R.transpose(-1, -2)
        th = tt_kpts - (Rh @ ss_kpts.unsqueeze(-1)).squeeze()
        tss = ss_kpts[None] @ Rh.transpose(-1, -2) + th[:, None]
        diffs = torch.sqrt(torch.sum((tss - tt_kpts[None]) ** 2, dim=-1))
        thr = torch.sqrt(torch.sum(ss_kpts ** 2, dim=-1)) * np.pi / cfg.patch.azi_n * cfg.match.inlier_th
        sign = diffs < thr[None]
        inlier_num = torch.sum(sign, dim=-1)
        best = torch.argmax(inlier_num)
        inlier_ind = torch.where(sign[best])[0]
        init = torch.eye(4)[None].clone()
        init[0, :3, :3] = Rh[best]
        init[0, :3, 3] = th[best]
        refined = model.post_refinement(init.clone(), ss_kpts[None], tt_kpts[None])
    np.savez_compressed(
        os.path.join(GOLD, 'match_tiny.npz'), src_desc=out['desc'].numpy(), tgt_desc=tout['desc'].numpy(),
        src_equi=out['equi'].numpy(), tgt_equi=tout['equi'].numpy(), src_R=out['R'].numpy(), tgt_R=tout['R'].numpy(),
        src_kpts=kpts.numpy(), tgt_kpts=tk.numpy(), s_mids=np.asarray(s_mids, np.int64),
        t_mids=np.asarray(t_mids, np.int64), ind=ind.numpy(), R_hyp=Rh.numpy(), t_hyp=th.numpy(),
        inlier_num=inlier_num.numpy(), best=int(best), inlier_ind=inlier_ind.numpy(), init_pose=init.numpy(),
        refined_pose=refined.numpy(), gt=gt.numpy())
    print('F5 matches', len(s_mids), 'best inliers', int(inlier_num[best]), 'refined vs gt err',
          float((refined[0] - gt).abs().max()))

    # ---- F6: Registration-Recall evaluator on synthetic gt.log / gt.info -----------------------------
    # (ThreeDMatch/test.py imports nibabel/open3d at module level; the stubs above satisfy that.)
    try:
        make_rr_fixture()
    except Exception as e:  # pragma: no cover
        print('F6 skipped:', repr(e))


def make_kitti_fixture():
    """F7 `kitti_tiny.npz`: the KITTI branch of the reference (KITTI/config.py constants, released KITTI weights, the
    R = I alignment of patch_embedder.py:143-147) on a small ring-pattern LiDAR pair: pyramid tables, EFCNN / DetNet
    outputs and the descriptors of 48 FPS keypoints."""
    import torch.nn.functional as F
    torch.manual_seed(0)
    np.random.seed(0)
    cfg, model, _ = load_reference_model('KITTI', '06050001')
    from KITTI import dataloader as dlk
    sample = synth.make_kitti_pair(9, beams=24, az_steps=700)
    limits = dlk.calibrate_neighbors([sample], cfg, dlk.collate_fn_descriptor)
    batch = dlk.collate_fn_descriptor([sample], cfg, limits)
    out = {'limits': np.asarray(limits, np.int32)}
    for l in range(3):
        out[f'points_{l}'] = batch['points'][l].numpy()
        out[f'neighbors_{l}'] = batch['neighbors'][l].numpy().astype(np.int32)
        out[f'pools_{l}'] = batch['pools'][l].numpy().astype(np.int32)
        out[f'upsamples_{l}'] = batch['upsamples'][l].numpy().astype(np.int32)
        out[f'lengths_{l}'] = batch['stack_lengths'][l].numpy().astype(np.int32)
    with torch.no_grad():
        axis, eps, branch = model.Ref(batch)
        bottle = branch['bottle_feature']
        score = model.Keypt(batch, {'bottle_feature': bottle, 'skip_feature': list(branch['skip_feature'])})
    out.update(features=batch['features'].numpy(), axis=axis.numpy(), eps=eps.numpy(), score=score.numpy())
    n_src = int(batch['stack_lengths'][0][0])
    src_pts = batch['src_pcd']
    src_axis = F.normalize(axis[:n_src], p=2, dim=1)
    mask = (torch.sum(-src_axis * src_pts, dim=1) < 0).float().unsqueeze(1)
    src_axis = src_axis * (1 - mask) - src_axis * mask
    fps_idx = torch.from_numpy(cpu.fps(src_pts[None].numpy(), 48)).long()[0]
    kpts, kaxis = src_pts[fps_idx], src_axis[fps_idx]
    raw = batch['src_pcd_raw']
    perm = np.random.RandomState(7).permutation(raw.shape[0])
    orig_choice = np.random.choice
    np.random.choice = lambda n, size=None, replace=True: perm           # pin select_patches' shuffle
    with torch.no_grad():
        d = model.Desc(raw[None], kpts[None], kaxis[None])
    np.random.choice = orig_choice
    out.update(raw=raw.numpy(), kpts=kpts.numpy(), kaxis=kaxis.numpy(), perm=perm.astype(np.int64),
               patches=d['patches'].numpy(), R=d['R'].numpy(), rand_axis=d['rand_axis'].numpy(), desc=d['desc'].numpy(),
               equi=d['equi'].numpy())
    np.savez_compressed(os.path.join(GOLD, 'kitti_tiny.npz'), **out)
    print('F7 kitti: layers', [p.shape[0] for p in batch['points']], 'limits', limits, 'desc', d['desc'].shape,
          'R==I', bool((d['R'] == torch.eye(3)).all()))
    sys.modules.pop('KITTI.config', None)


def name_of(dataset):
    return 'kitti_full_1500.npz' if dataset == 'kitti' else 'full_1500.npz'


def make_full_fixture(seed=4242, P=1500, dataset='3dmatch'):
    """F8 `full_1500.npz` (round 5): ONE synth.make_pair at the full 3DMatch shape through the reference's OWN
    buffer.forward (models/BUFFER.py:231-333, unmodified) at the reference's 1500 keypoints: the collate of
    ThreeDMatch/dataloader.py, Ref, Keypt, FPS, Desc x 2, mutual matching, Inlier, hypotheses, RANSAC, post_refinement.
    Third-party operators are the stubs named at the top of this file; open3d's RANSAC is replaced by the numpy restatement of
    the PRODUCT's deterministic sampler (oracle/pipeline_ref.ransac_kabsch, seed = the pair seed) so that the refined pose of
    the reference's forward is comparable with the device pose.  Stored: the seed (inputs are regenerated by synth.make_pair;
    float64 checksums guard the generator), neighbour limits, layer sizes, every keypoint index, match ids, `ind`, inlier
    counts, the winner and its inlier list, RANSAC and refined poses, sampled rows of desc / equi / axis / score and float64
    checksums of the full tensors."""
    torch.manual_seed(0)
    np.random.seed(0)
    from oracle import pipeline_ref
    if dataset == 'kitti':                                   # F9: the KITTI branch (KITTI/config.py, released KITTI snapshot, R = I alignment, no refinement)
        cfg, model, _ = load_reference_model('KITTI', '06050001')
        from KITTI import dataloader as dl
        sample = synth.make_kitti_pair(seed)
    else:
        from ThreeDMatch import dataloader as dl
        cfg, model, _ = load_reference_model()
        sample = synth.make_pair(seed)
    assert cfg.point.num_keypts == P
    limits = dl.calibrate_neighbors([sample], cfg, dl.collate_fn_descriptor)
    batch = dl.collate_fn_descriptor([sample], cfg, limits)
    rng = np.random.default_rng(seed)
    perms = [rng.permutation(len(sample['src_fds_pts'])), rng.permutation(len(sample['tgt_fds_pts']))]
    cap = dict(fps=[], desc=[], desc_kpts=[], perm_calls=0)

    # --- open3d surface of models/BUFFER.py:314-326 and utils/common.py:569-578 (third-party: stubbed) ---
    class _PC:
        points = None
        colors = None

    def ransac(pcd0, pcd1, corr, max_dist, est, n, checkers, criteria):
        c = np.asarray(corr)
        T, info = pipeline_ref.ransac_kabsch(np.asarray(pcd0.points), np.asarray(pcd1.points), c[:, 0], 4096, seed, max_dist,
                                             cfg.match.similar_th)
        cap.update(corr=c.copy(), init_pose=np.asarray(T, np.float64), ransac_info=info)
        return types.SimpleNamespace(transformation=np.asarray(T, np.float64))

    ident = lambda *a, **k: (a[0] if a else None)
    reg = types.SimpleNamespace(registration_ransac_based_on_correspondence=ransac, TransformationEstimationPointToPoint=ident,
                                CorrespondenceCheckerBasedOnEdgeLength=ident, CorrespondenceCheckerBasedOnDistance=ident,
                                RANSACConvergenceCriteria=lambda *a: a)
    o3d = types.SimpleNamespace(geometry=types.SimpleNamespace(PointCloud=_PC),
                                utility=types.SimpleNamespace(Vector3dVector=lambda x: np.asarray(x, np.float64),
                                                              Vector2iVector=lambda x: np.asarray(x, np.int64)),
                                pipelines=types.SimpleNamespace(registration=reg))
    import models.BUFFER as MB
    import utils.common as UC
    MB.o3d = o3d
    UC.open3d = o3d

    # --- spies: FPS inputs / outputs, the two Desc calls, the pinned shuffles of select_patches ---
    pu = sys.modules['pointnet2_ops.pointnet2_utils']
    orig_fps = pu.furthest_point_sample

    def spy_fps(xyz, m):
        idx = orig_fps(xyz, m)
        cap['fps'].append((xyz[0].numpy().copy(), idx[0].numpy().copy()))
        return idx

    pu.furthest_point_sample = spy_fps
    MB.pnt2.furthest_point_sample = spy_fps
    orig_choice = np.random.choice

    def pinned_choice(n, size=None, replace=True):
        p = perms[cap['perm_calls']]
        cap['perm_calls'] += 1
        assert len(p) == n
        return p

    np.random.choice = pinned_choice
    orig_desc = model.Desc.forward

    def spy_desc(*a, **k):
        out = orig_desc(*a, **k)
        cap['desc_kpts'].append(a[1][0].numpy().copy())                     # the keypoints Desc actually received ([1,P,3])
        cap['desc'].append({kk: v.numpy().copy() for kk, v in out.items() if isinstance(v, torch.Tensor)})
        return out

    model.Desc.forward = spy_desc
    orig_mm = model.mutual_matching

    def spy_mm(a, b):
        s_mids, t_mids = orig_mm(a, b)
        cap['s_mids'], cap['t_mids'] = np.asarray(s_mids, np.int64), np.asarray(t_mids, np.int64)
        return s_mids, t_mids

    model.mutual_matching = spy_mm
    hook = model.Inlier.register_forward_hook(lambda m, i, o: cap.update(ind=o.numpy().copy()))
    import time
    t0 = time.time()
    with torch.no_grad():
        pose, src_axis, tgt_axis = model(batch)
        axis, eps, branch = model.Ref(batch)              # (again, for the sampled rows: forward does not return them)
        score = model.Keypt(batch, branch)
    print('F8 reference forward', round(time.time() - t0, 1), 's')
    hook.remove()
    np.random.choice = orig_choice
    pu.furthest_point_sample = orig_fps
    MB.pnt2.furthest_point_sample = orig_fps
    model.Desc.forward = orig_desc
    model.mutual_matching = orig_mm
    assert cap['perm_calls'] == 2 and len(cap['fps']) == 2 and len(cap['desc']) == 2

    n_src = int(batch['stack_lengths'][0][0])
    sc = score[:, 0].numpy()
    det = [np.nonzero(sc[:n_src] > cfg.point.keypts_th)[0], np.nonzero(sc[n_src:] > cfg.point.keypts_th)[0]]
    kp_idx = [det[i][cap['fps'][i][1].astype(np.int64)] for i in range(2)]            # rows of the src / tgt halves of points[0]
    pts0 = batch['points'][0].numpy()
    for i, off in enumerate((0, n_src)):
        # the reconstructed indices (det[fps idx]) ARE the keypoints the reference's Desc received (this used to compare a
        # value with itself)
        assert np.array_equal(pts0[off + kp_idx[i]], cap['desc_kpts'][i]), f'reconstructed keypoints of cloud {i} differ from those Desc received'
    # hypotheses / scoring as the forward computed them (recomputed here from the captured pieces: forward keeps them local)
    s_mids, t_mids, ind = cap['s_mids'], cap['t_mids'], cap['ind']
    kp = [torch.from_numpy(pts0[kp_idx[0]]), torch.from_numpy(pts0[n_src + kp_idx[1]])]
    ss, tt = kp[0][s_mids], kp[1][t_mids]
    import kornia.geometry.conversions as Convert
    angle = torch.from_numpy(ind) * 2 * np.pi / cfg.patch.azi_n + 1e-6
    aa = torch.zeros_like(ss)
    aa[:, -1] = 1
    azi_R = Convert.angle_axis_to_rotation_matrix(aa * angle[:, None])
    Rh = torch.from_numpy(cap['desc'][1]['R'])[t_mids] @ azi_R @ torch.from_numpy(cap['desc'][0]['R'])[s_mids].transpose(-1, -2)
    th = tt - (Rh @ ss.unsqueeze(-1)).squeeze()
    diffs = torch.sqrt(torch.sum((ss[None] @ Rh.transpose(-1, -2) + th[:, None] - tt[None]) ** 2, dim=-1))
    thr = torch.sqrt(torch.sum(ss ** 2, dim=-1)) * np.pi / cfg.patch.azi_n * cfg.match.inlier_th
    inlier_num = torch.sum(diffs < thr[None], dim=-1)
    best = int(torch.argmax(inlier_num))
    assert np.array_equal(np.nonzero((diffs < thr[None])[best].numpy())[0], cap['corr'][:, 0])

    r = np.random.default_rng(1)
    rows_p = np.sort(r.choice(P, 256, replace=False))           # sampled keypoint rows (desc, R)
    rows_e = rows_p[:48]                                        # ... of them for the 17.9 KB equivariant maps
    rows_n = np.sort(r.choice(pts0.shape[0], 1024, replace=False))
    f64 = lambda a: np.array([np.asarray(a, np.float64).sum(), np.abs(np.asarray(a, np.float64)).sum()])
    out = dict(seed=seed, num_keypts=P, limits=np.asarray(limits, np.int32),
               layer_sizes=np.array([p.shape[0] for p in batch['points']], np.int32), n_src=n_src,
               in_checksums=np.stack([f64(sample[k]) for k in ('src_fds_pts', 'tgt_fds_pts', 'src_sds_pts', 'tgt_sds_pts')]),
               in_shapes=np.array([sample[k].shape[0] for k in ('src_fds_pts', 'tgt_fds_pts', 'src_sds_pts', 'tgt_sds_pts')]),
               relt_pose=np.asarray(sample['relt_pose'], np.float64),
               kp_idx_src=kp_idx[0].astype(np.int32), kp_idx_tgt=kp_idx[1].astype(np.int32),
               n_candidates=np.array([len(d) for d in det], np.int32),
               s_mids=s_mids, t_mids=t_mids, ind=ind, inlier_num=inlier_num.numpy().astype(np.int32), best=best,
               inlier_ind=cap['corr'][:, 0].astype(np.int32), init_pose=cap['init_pose'], pose=np.asarray(pose, np.float64),
               ransac_info=np.array(cap['ransac_info'], np.int64),
               rows_p=rows_p, rows_e=rows_e, rows_n=rows_n,
               axis_rows=axis.numpy()[rows_n], eps_rows=eps.numpy()[rows_n], score_rows=score.numpy()[rows_n],
               axis_sum=f64(axis.numpy()), eps_sum=f64(eps.numpy()), score_sum=f64(score.numpy()))
    for i, nm in enumerate(('src', 'tgt')):
        d = cap['desc'][i]
        out.update({f'{nm}_desc_rows': d['desc'][rows_p], f'{nm}_equi_rows': d['equi'][rows_e], f'{nm}_R_rows': d['R'][rows_p],
                    f'{nm}_rand_axis_rows': d['rand_axis'][rows_p], f'{nm}_desc_sum': f64(d['desc']), f'{nm}_equi_sum': f64(d['equi']),
                    f'{nm}_patches_sum': f64(d['patches']), f'{nm}_equi_rowsum': np.asarray(d['equi'], np.float64).sum((1, 2, 3))})
    # Rows whose voxel ball queries hinge on the last bits of the aligned coordinates (buffer_amd/diagnose.py: a point of the patch
    # within the alignment's last-bit uncertainty of a ball surface): the ONLY rows of another correct fp32 implementation that may
    # legitimately differ from these by more than round-off.  Their aligned patches and full outputs ride along, so the test can show
    # the cause row by row (round 6).
    from buffer_amd import diagnose
    from oracle import torch_ref as T_
    centres = T_.voxel_centres(cfg.patch.rad_n, cfg.patch.azi_n, cfg.patch.ele_n).numpy()
    voxel_r = cfg.patch.delta / cfg.patch.rad_n
    for i, nm in enumerate(('src', 'tgt')):
        d = cap['desc'][i]
        rows = np.array([r for r in range(P) if diagnose.explain_row(d['patches'][r], centres, voxel_r)['near_surface_pairs'] > 0], np.int64)
        out.update({f'{nm}_surface_rows': rows, f'{nm}_surface_patches': d['patches'][rows], f'{nm}_surface_desc': d['desc'][rows],
                    f'{nm}_surface_equi_rowsum': np.asarray(d['equi'][rows], np.float64).sum((1, 2, 3))})
        print(f'{name_of(dataset)} {nm}: {len(rows)} of {P} patches have a point on a voxel ball surface (to the last bits of the alignment)')
    name = name_of(dataset)
    np.savez_compressed(os.path.join(os.environ.get('BUF_GOLDEN_OUT', GOLD), name), **out)
    err = np.abs(np.asarray(pose) - sample['relt_pose']).max()
    print(name, ': layers', out['layer_sizes'], 'limits', limits, 'candidates', out['n_candidates'], 'matches', len(s_mids),
          'best inliers', int(inlier_num[best]), 'ransac', cap['ransac_info'], '|pose - gt|', float(err),
          'size', os.path.getsize(os.path.join(GOLD, name)))
    if dataset == 'kitti':
        sys.modules.pop('KITTI.config', None)


def make_rr_fixture():
    src = open(f'{REF}/ThreeDMatch/test.py').read()
    # only the function definitions above `if __name__` are needed; exec them in a module namespace
    head = src.split("if __name__ == '__main__':")[0]
    head = head.replace('os.environ["CUDA_VISIBLE_DEVICES"]', '_ignored')
    ns = {}
    exec(compile(head, 'ref_test_head', 'exec'), ns)
    rng = np.random.default_rng(3)
    n_frag = 12
    pairs, gt_T, est_T, infos = [], [], [], []
    for i in range(n_frag):
        for j in range(i + 1, n_frag):
            if rng.random() < 0.35:
                R = synth.random_rotation(rng)
                T = np.eye(4)
                T[:3, :3] = R
                T[:3, 3] = rng.normal(size=3)
                pairs.append((i, j))
                gt_T.append(T)
                A = rng.normal(size=(6, 6))
                infos.append(A @ A.T * 50)
                E = T.copy()
                if rng.random() < 0.3:
                    E[:3, 3] += rng.normal(scale=0.5, size=3)
                else:
                    E[:3, 3] += rng.normal(scale=0.01, size=3)
                est_T.append(E)
    import tempfile
    d = tempfile.mkdtemp()

    def write_log(path, Ts):
        with open(path, 'w') as f:
            for (i, j), T in zip(pairs, Ts):
                f.write(f'{i}\t {j}\t {n_frag}\n')
                for r in range(4):
                    f.write('\t'.join(f'{v:.8e}' for v in T[r]) + '\n')

    write_log(os.path.join(d, 'gt.log'), gt_T)
    write_log(os.path.join(d, 'est.log'), est_T)
    with open(os.path.join(d, 'gt.info'), 'w') as f:
        for (i, j), I in zip(pairs, infos):
            f.write(f'{i}\t {j}\t {n_frag}\n')
            for r in range(6):
                f.write('\t'.join(f'{v:.8e}' for v in I[r]) + '\n')
    gt_pairs, gt_traj = ns['read_trajectory'](os.path.join(d, 'gt.log'))
    n_fragments, gt_traj_cov = ns['read_trajectory_info'](os.path.join(d, 'gt.info'))
    est_pairs, est_traj = ns['read_trajectory'](os.path.join(d, 'est.log'))
    res = ns['evaluate_registration'](n_fragments, est_traj, est_pairs, gt_pairs, gt_traj, gt_traj_cov)
    texts = {k: open(os.path.join(d, k)).read() for k in ('gt.log', 'est.log', 'gt.info')}
    flat = np.array([res[0], res[1]], np.float64)
    np.savez_compressed(os.path.join(GOLD, 'rr_eval.npz'), gt_log=texts['gt.log'], est_log=texts['est.log'],
                        gt_info=texts['gt.info'], result=flat, flags=np.asarray(res[2], np.int32),
                        errors=np.asarray(res[3], np.float64), n_fragments=n_frag,
                        pairs=np.asarray(pairs, np.int32), gt_T=np.asarray(gt_T), est_T=np.asarray(est_T),
                        infos=np.asarray(infos))
    print('F6 evaluate_registration ->', flat)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'kitti':       # only F7 (leaves F1-F6 untouched)
    cpu.build(ref=True)
    install_stubs()
    os.makedirs(GOLD, exist_ok=True)
    make_kitti_fixture()
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] in ('full', 'kitti_full'):     # only F8 / F9 (leaves F1-F7 untouched)
    cpu.build(ref=True)
    install_stubs()
    if sys.argv[1] == 'full':
        make_full_fixture()
    else:
        make_full_fixture(seed=2003, dataset='kitti')
    sys.exit(0)

if __name__ == '__main__':
    main()
