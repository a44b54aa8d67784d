"""End-to-end registration of one fragment pair on one GPU (the inference branch of buffer.forward,
models/BUFFER.py:231-333, with the collate stage of ThreeDMatch/dataloader.py:115-245 moved on device)."""
import numpy as np
import torch

from . import ops, pyramid, registration
from .config import THREEDMATCH
from .patch_embedder import PatchEmbedder
from .point_learner import PointLearner, orient_axes
from .weights import load_weights


class BufferPipeline:
    def __init__(self, cfg=THREEDMATCH, device='cuda:0', weights=None, limits=None):
        self.cfg, self.device = cfg, torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('BufferPipeline runs on a HIP device only (no CPU path)')
        W = weights if weights is not None else load_weights(cfg.weights)
        self.W = W
        self.point = PointLearner(W, self.device, cfg.scale)
        self.desc = PatchEmbedder(W, self.device, cfg)
        self.inlier = registration.CostVolume(W, self.device, cfg.azi_n)
        self.limits = None if limits is None else [int(x) for x in limits]

    def calibrate(self, samples):
        self.limits = [int(x) for x in pyramid.calibrate_limits(samples, self.cfg, self.device)]
        return self.limits

    def upload(self, sample):
        """host sample dict -> device-resident inputs (what the timed region of bench.py starts from)."""
        pts, lens, feats, src_raw, tgt_raw = pyramid.stack_sample(sample, self.device)
        return dict(points=pts, lengths=lens, features=feats, src_raw=src_raw, tgt_raw=tgt_raw)

    @torch.no_grad()
    def register(self, inp, seed=0, perms=None, detail=False):
        """inp from upload() -> pose f32[4,4] (src -> tgt), device tensor."""
        cfg = self.cfg
        if self.limits is None:
            raise RuntimeError('neighbourhood limits not calibrated: call calibrate() or pass limits=')
        pyr = pyramid.build_pyramid(inp['points'], inp['lengths'], self.limits, cfg)
        n_src = int(inp['lengths'][0])
        axis, eps, bottle, skips, _ = self.point.efcnn(pyr, inp['features'])
        score = self.point.detnet(pyr, bottle, skips)
        pts0 = pyr['points'][0]
        out = {}
        cand_p, cand_a = [], []
        for lo, hi in ((0, n_src), (n_src, pts0.shape[0])):
            p = pts0[lo:hi]
            a = orient_axes(axis[lo:hi], p)
            keep = torch.nonzero(score[lo:hi, 0] > cfg.keypts_th).flatten()         # BUFFER.py:255-259
            if keep.shape[0] == 0:
                return self._identity(out, detail)
            cand_p.append(p[keep]); cand_a.append(a[keep])
        # both fragments sampled in one launch, one workgroup per cloud (BUFFER.py:266-271)
        fps = ops.furthest_point_sample_ragged(torch.cat(cand_p), [c.shape[0] for c in cand_p], cfg.num_keypts).long()
        kp = [cand_p[i][fps[i]].contiguous() for i in range(2)]
        ka = [cand_a[i][fps[i]].contiguous() for i in range(2)]
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        res = []
        for i, raw in enumerate((inp['src_raw'], inp['tgt_raw'])):
            perm = perms[i] if perms is not None else torch.randperm(raw.shape[0], device=self.device, generator=g)
            res.append(self.desc(raw, kp[i], ka[i], perm))
        s_mids, t_mids = registration.mutual_matching(res[0]['desc'], res[1]['desc'])
        if s_mids.shape[0] < 3:
            return self._identity(out, detail)
        ss_kpts, tt_kpts = kp[0][s_mids].contiguous(), kp[1][t_mids].contiguous()
        e = cfg.ele_n
        ind = self.inlier(res[0]['equi'][s_mids][:, :, 1:e - 1].contiguous(),
                          res[1]['equi'][t_mids][:, :, 1:e - 1].contiguous())
        pose, diag = registration.recover_pose(ind, ss_kpts, tt_kpts, res[0]['R'][s_mids].contiguous(),
                                               res[1]['R'][t_mids].contiguous(), cfg, seed)
        if detail:
            out.update(dict(pyr=pyr, axis=axis, eps=eps, score=score, kpts=kp, kaxis=ka, desc=res, s_mids=s_mids,
                            t_mids=t_mids, ind=ind, **diag))
            return pose, out
        return pose

    def _identity(self, out, detail):
        pose = torch.eye(4, device=self.device)       # ThreeDMatch/test.py:242-245: failed pair -> identity
        return (pose, out) if detail else pose
