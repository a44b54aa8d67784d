"""End-to-end registration of one fragment pair on one GPU (the inference branch of buffer.forward,
models/BUFFER.py:231-333, with the collate stage of ThreeDMatch/dataloader.py:115-245 moved on device)."""
import os
import time

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
        self.inlier = registration.CostVolume(W, self.device, cfg.azi_n, getattr(cfg, 'cnn_arith', 'f32'))
        self.limits = None if limits is None else [int(x) for x in limits]
        self.host_wait_s = 0.0        # seconds the enqueueing thread spent blocked in the path's host round trips (diagnostics)

    def check_range(self):
        """cnn_arith='split': the kernels are safe by construction (csrc/split_safe.hip: every patch / match whose values leave the f16
        range is recomputed by the fp32 kernel in the same stream), so there is nothing to check and no synchronisation here.  Only a
        split network without an fp32 kernel for its widths (never the released ones) keeps the round-5 contract: FloatingPointError
        if its status word was set."""
        for m in (self.desc.fused, self.inlier.fused):
            if hasattr(m, 'check_range') and not getattr(m, 'safe', False):
                m.check_range()

    def range_fallbacks(self):
        """(patches, matches) of the LAST launches that left the f16 range and took the fp32 kernels (cnn_arith='split'; synchronises)"""
        return tuple(m.range_fallbacks() if hasattr(m, 'range_fallbacks') else 0 for m in (self.desc.fused, self.inlier.fused))

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
            keep = ops.compact_greater(score[lo:hi, 0], cfg.keypts_th).long()       # BUFFER.py:255-259
            if keep.shape[0] == 0:
                return self._identity(out, detail)
            cand_p.append(p[keep]); cand_a.append(a[keep])
        # both fragments sampled in one launch, one workgroup per cloud (BUFFER.py:266-271)
        fps = ops.furthest_point_sample_ragged(torch.cat(cand_p), [c.shape[0] for c in cand_p], cfg.num_keypts).long()
        kp = [cand_p[i][fps[i]].contiguous() for i in range(2)]
        ka = [cand_a[i][fps[i]].contiguous() for i in range(2)]
        raws = (inp['src_raw'], inp['tgt_raw'])
        if perms is not None:                               # caller-pinned permutations (parity tests)
            sup = torch.cat([raws[i][perms[i]] for i in range(2)]).contiguous()
            sup_len = [raws[0].shape[0], raws[1].shape[0]]
        else:                                               # keyed device permutation, the same one register_batch uses
            sup, sup_len = ops.permute_clouds(raws, [ops.perm_key(seed, j) for j in range(2)])
        P = cfg.num_keypts
        patches = ops.select_patches_batched(sup, sup_len, torch.cat(kp), P, cfg.des_r, cfg.num_points_per_patch)
        emb = self.desc.embed_patches(patches, torch.cat(ka), want_patches=detail)
        res = [{k: (v[i * P:(i + 1) * P] if v is not None else None) for k, v in emb.items()} for i in range(2)]
        s_mids, t_mids = registration.mutual_matching(res[0]['desc'], res[1]['desc'])
        if s_mids.shape[0] < 3:
            return self._identity(out, detail)
        ss_kpts, tt_kpts = kp[0][s_mids].contiguous(), kp[1][t_mids].contiguous()
        e = cfg.ele_n
        ind = self.inlier(res[0]['equi'][s_mids][:, :, 1:e - 1].contiguous(),
                          res[1]['equi'][t_mids][:, :, 1:e - 1].contiguous())
        pose, diag = registration.recover_pose(ind, ss_kpts, tt_kpts, res[0]['R'][s_mids].contiguous(),
                                               res[1]['R'][t_mids].contiguous(), cfg, seed)
        self.check_range()
        if detail:
            out.update(dict(pyr=pyr, axis=axis, eps=eps, score=score, kpts=kp, kaxis=ka, desc=res, s_mids=s_mids,
                            t_mids=t_mids, ind=ind, **diag))
            return pose, out
        return pose

    @torch.no_grad()
    def register_batch(self, inps, seeds=None, perms=None):
        """Several pairs through ONE set of launches per stage (the MI355X-native form: the pyramid, the VN
        blocks, FPS (one workgroup per cloud), patch selection, voxelisation, both CNNs and the 1-NN search all take the
        stacked batch; only the per-pair pose recovery loops).  inps: list of upload() dicts ->
        list of pose f32[4,4] device tensors.  Per pair the arithmetic is that of register()."""
        poses = self._describe_and_match(self._keypoints(inps, seeds, perms))
        self.check_range()
        return poses

    @torch.no_grad()
    def register_batches(self, batches, seeds=None):
        """A sequence of batches, software-pipelined over two HIP streams: the keypoint stage of batch i+1 (pyramid, point
        learner, FPS -- short kernels, FPS latency-bound on 2B of the 256 CUs) is enqueued on a high-priority side stream
        BEFORE the descriptor / matching stage of batch i goes onto the current stream, so it runs beside the chip-filling
        CNN kernels instead of in front of them.  Results are those of register_batch batch by batch.
        batches: list of lists of upload() dicts -- or of callables returning such a list, which are then evaluated on the side
        stream as part of the keypoint stage (device pre-processing of the next batch beside the CNN kernels of this one);
        seeds: list of lists -> list of lists of poses."""
        dev = self.device
        main = torch.cuda.current_stream(dev)
        if not hasattr(self, '_kp_stream'):
            self._kp_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get('BUF_KP_STREAM_PRIORITY', -1)))
        side = self._kp_stream
        seeds = [None] * len(batches) if seeds is None else seeds
        out = []

        def stage1(i):
            with torch.cuda.stream(side):
                inps = batches[i]() if callable(batches[i]) else batches[i]
                st = self._keypoints(inps, seeds[i], None)
                if callable(batches[i]):                   # inputs made on the side stream are read on the current one too
                    st['cross'] = tuple(st.get('cross', ())) + tuple(v for x in inps for v in x.values() if isinstance(v, torch.Tensor))
                ev = torch.cuda.Event()
                ev.record(side)
            return st, ev

        side.wait_stream(main)
        nxt = stage1(0) if batches else None
        for i in range(len(batches)):
            st, ev = nxt
            main.wait_event(ev)
            for t in st.get('cross', ()):                  # produced on the side stream, consumed on the current one
                t.record_stream(main)
            st = self._describe(st)                        # ~300 ms of CNN work queued on the current stream, no host sync
            # the next batch's keypoint stage goes out NOW: its host round trip (per-cloud candidate counts) waits on the
            # side stream only, while the current stream is busy with the kernels queued above
            nxt = stage1(i + 1) if i + 1 < len(batches) else None
            out.append(self._match(st))
        self.check_range()
        return out

    def _keypoints(self, inps, seeds, perms):
        """pyramid -> point learner -> threshold -> FPS for a stacked batch -> state for _describe_and_match."""
        cfg, dev = self.cfg, self.device
        B = len(inps)
        if self.limits is None:
            raise RuntimeError('neighbourhood limits not calibrated: call calibrate() or pass limits=')
        seeds = list(range(B)) if seeds is None else list(seeds)
        lens = np.concatenate([np.asarray(i['lengths'], np.int32) for i in inps])            # [2B]
        pts = torch.cat([i['points'] for i in inps]) if B > 1 else inps[0]['points']
        feats = torch.cat([i['features'] for i in inps]) if B > 1 else inps[0]['features']
        w0 = ops.HOST_WAIT_S[0]
        pyr = pyramid.build_pyramid(pts, lens, self.limits, cfg)
        self.host_wait_s += ops.HOST_WAIT_S[0] - w0                                         # the two subsample row-count round trips
        pair_rows = lens.reshape(B, 2).sum(1)
        seg = pair_rows.astype(np.int32) if B > 1 else None      # InstanceNorm segments of the score heads: one per pair
        axis, eps, bottle, skips, _ = self.point.efcnn(pyr, feats, seg)
        score = self.point.detnet(pyr, bottle, skips, seg)
        pts0 = pyr['points'][0]
        cloud_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
        cloud_id = torch.repeat_interleave(torch.arange(2 * B, device=dev), cloud_len)
        axis_o = orient_axes(axis, pts0)                                                    # BUFFER.py:244-249 (row-wise)
        t0 = time.perf_counter()                                                            # host round trip 1: the compaction sizes its
        keep = ops.compact_greater(score[:, 0], cfg.keypts_th).long()                        # output (:255-259, ascending), then the
        counts = torch.bincount(cloud_id[keep], minlength=2 * B).cpu().numpy()               # candidate counts per cloud for FPS
        self.host_wait_s += time.perf_counter() - t0
        st = dict(inps=inps, seeds=seeds, perms=perms, B=B)
        if (counts == 0).any():                             # rare: some cloud has no point above the threshold
            st['starved'] = set(int(c) // 2 for c in np.nonzero(counts == 0)[0])
            return st
        cand_p, cand_a = pts0[keep].contiguous(), axis_o[keep].contiguous()
        fps = ops.furthest_point_sample_ragged(cand_p, counts, cfg.num_keypts).long()       # one workgroup per cloud
        off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)).to(dev)
        gidx = (fps + off[:, None]).reshape(-1)
        st['kp'], st['ka'] = cand_p[gidx].contiguous(), cand_a[gidx].contiguous()           # [2B*P, 3]
        st['cross'] = (st['kp'], st['ka'])
        return st

    def _describe_and_match(self, st):
        """patches -> descriptors -> mutual matches -> cost volume -> per-pair pose recovery."""
        return self._match(self._describe(st))

    def _describe(self, st):
        """patch selection, voxelisation, descriptor CNN and the two 1-NN searches of a stacked batch: everything up to
        the first host round trip of the stage (the match count), enqueued without blocking."""
        cfg, dev = self.cfg, self.device
        inps, seeds, perms, B = st['inps'], st['seeds'], st['perms'], st['B']
        poses = [None] * B
        if 'starved' in st:
            bad = st['starved']
            for b in (b for b in range(B) if b not in bad):        # redo the healthy pairs one by one
                poses[b] = self.register(inps[b], seed=seeds[b], perms=perms[b] if perms is not None else None)
            st['poses'] = [p if p is not None else torch.eye(4, device=dev) for p in poses]
            return st
        kp, ka = st['kp'], st['ka']
        P = cfg.num_keypts
        raws = [r for i in inps for r in (i['src_raw'], i['tgt_raw'])]
        if perms is not None:
            sup = torch.cat([raws[2 * b + j][perms[b][j]] for b in range(B) for j in range(2)]).contiguous()
            sup_len = [r.shape[0] for r in raws]
        else:                                               # one launch shuffles every cloud of the step (keyed per pair seed)
            sup, sup_len = ops.permute_clouds(raws, [ops.perm_key(seeds[b], j) for b in range(B) for j in range(2)])
        patches = ops.select_patches_batched(sup, sup_len, kp, P, cfg.des_r, cfg.num_points_per_patch)   # one grid, one launch
        emb = self.desc.embed_patches(patches, ka)
        desc = emb['desc'].view(B, 2, P, -1)
        _, s_idx = ops.knn(desc[:, 1].contiguous(), desc[:, 0].contiguous(), 1)            # BUFFER.py:347: ref = tgt
        _, t_idx = ops.knn(desc[:, 0].contiguous(), desc[:, 1].contiguous(), 1)
        s_nn, t_nn = s_idx[:, :, 0], t_idx[:, :, 0]
        st['mutual'] = t_nn.gather(1, s_nn) == torch.arange(P, device=dev)[None]
        st['s_nn'], st['emb'] = s_nn, emb
        return st

    def _match(self, st):
        """mutual matches (first host round trip) -> cost volume -> per-pair pose recovery."""
        if 'poses' in st:
            return st['poses']
        cfg, dev = self.cfg, self.device
        seeds, B, kp, emb, s_nn, P = st['seeds'], st['B'], st['kp'], st['emb'], st['s_nn'], self.cfg.num_keypts
        poses = [None] * B
        t0 = time.perf_counter()                                                            # host round trip 2: the mutual matches
        mm = torch.nonzero(st['mutual'])                                                    # (pair, s) ascending; sizes its output
        pair_of, s_mid = mm[:, 0], mm[:, 1]
        t_mid = s_nn[pair_of, s_mid]
        m_counts = torch.bincount(pair_of, minlength=B).cpu().numpy()                       # matches per pair
        self.host_wait_s += time.perf_counter() - t0
        src_row = (2 * pair_of) * P + s_mid
        tgt_row = (2 * pair_of + 1) * P + t_mid
        ind = self.inlier.gathered(emb['equi'], src_row, tgt_row)      # BUFFER.py:291-292 rows 1..ele_n-2, gathered in-kernel
        ss_all, tt_all = kp[src_row].contiguous(), kp[tgt_row].contiguous()
        sR_all, tR_all = emb['R'][src_row].contiguous(), emb['R'][tgt_row].contiguous()
        # hypotheses, all-vs-all scoring, RANSAC and refinement of all B pairs: one set of launches (csrc/registration.hip,
        # batched section), bit-identical to the pair-by-pair recover_pose of register()
        all_poses = ops.recover_poses_batched(ind, ss_all, tt_all, sR_all, tR_all, m_counts, seeds, cfg)
        return [all_poses[b] for b in range(B)]

    def _identity(self, out, detail):
        pose = torch.eye(4, device=self.device)       # ThreeDMatch/test.py:242-245: failed pair -> identity
        return (pose, out) if detail else pose
