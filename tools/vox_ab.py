"""Development aid: bitwise A/B of k_patch_voxelize between two builds of the library.
    python tools/vox_ab.py gen OUT.npz      (library = BUF_LIB_PATH or the in-tree one)
    python tools/vox_ab.py cmp A.npz B.npz
Inputs: (a) 3DMatch-shaped patches cut from a synthetic room (select_patches), (b) uniform points in the unit ball, (c) adversarial
patches: points ON the ball surfaces, centre + r u (1 + delta), delta in {0, +-1e-7, +-3e-7, +-1e-6, +-1e-5} -- the pairs the
split-f16 distance form cannot call and hands to the reference's fp32 test."""
import sys
import time
import numpy as np


def gen(out):
    import torch
    from buffer_amd import ops, synth
    from buffer_amd.weights import load_weights
    from oracle import torch_ref as T          # (tool only: voxel centre table)
    dev = torch.device('cuda:0')
    W = load_weights('3dmatch')
    Wt = {k: torch.from_numpy(v) for k, v in W.items()}
    s = (Wt['Desc.pnt_layer.1.weight'] / torch.sqrt(Wt['Desc.pnt_layer.1.running_var'] + 1e-5)).numpy()
    t = (Wt['Desc.pnt_layer.1.bias'] - Wt['Desc.pnt_layer.1.running_mean'] * torch.from_numpy(s)).numpy()
    res = {}
    for name, (rad_n, azi_n, ele_n, nsample, npts) in dict(m3=(3, 20, 7, 10, 512), odd=(2, 8, 4, 16, 200)).items():
        centres = T.voxel_centres(rad_n, azi_n, ele_n).float()
        r = 0.8 / rad_n
        ang = -torch.arange(azi_n, dtype=torch.float64) * 2 * np.pi / azi_n
        azi_cs = torch.stack([torch.cos(ang), torch.sin(ang)], 1).float()
        g = torch.Generator().manual_seed(5)
        sets = {}
        # (b) uniform in the ball, keypoint (origin) last
        P = 6000
        u = torch.randn((P, npts, 3), generator=g)
        u = u / u.norm(dim=-1, keepdim=True) * torch.rand((P, npts, 1), generator=g) ** (1 / 3)
        u[:, -1] = 0
        sets['ball'] = u
        # (c) adversarial
        P = 3000
        ci = torch.randint(0, centres.shape[0], (P, npts), generator=g)
        d = torch.randn((P, npts, 3), generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        deltas = torch.tensor([0, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 1e-5, -1e-5], dtype=torch.float64)
        dl = deltas[torch.randint(0, len(deltas), (P, npts), generator=g)]
        a = (centres[ci].double() + r * d.double() * (1 + dl[..., None])).float()
        a[:, -1] = 0
        sets['surface'] = a
        if name == 'm3':                     # (a) real patch geometry
            sample = synth.make_pair(3, n_raw=60_000)
            raw = torch.from_numpy(sample['src_fds_pts']).float().to(dev)
            kp = raw[torch.randperm(raw.shape[0], generator=g)[:8000].to(dev)].contiguous()
            pt = ops.select_patches(raw, kp, 0.3, npts)
            sets['room'] = pt.cpu()
        for sname, patches in sets.items():
            des_r = 0.3 if sname == 'room' else 1.0
            ax = torch.nn.functional.normalize(torch.randn((patches.shape[0], 3), generator=g), dim=1) if sname == 'room' else None
            pd = patches.to(dev).contiguous()
            args = (pd, None if ax is None else ax.to(dev), des_r, centres.to(dev), azi_cs.to(dev), r, nsample,
                    W['Desc.pnt_layer.0.weight'].reshape(16, 3), W['Desc.pnt_layer.0.bias'], s, t, azi_n, True)
            x, R, ra, pn = ops.patch_voxelize(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ops.patch_voxelize(*args)
            torch.cuda.synchronize()
            print(f'{name}/{sname}: {patches.shape[0]} patches, {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call')
            res[f'{name}_{sname}'] = x.cpu().numpy()
    np.savez(out, **res)


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k].view(np.uint32), B[k].view(np.uint32))
        nd = int((A[k].view(np.uint32) != B[k].view(np.uint32)).sum())
        print(f'{k}: {A[k].shape} bitwise equal: {same} ({nd} differing values, max |diff| {np.abs(A[k] - B[k]).max():.3g})')
        bad += nd
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    sys.path.insert(0, '.')
    if sys.argv[1] == 'gen':
        gen(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
