"""Kernels of the path beside each other on two HIP streams (round 5).

Found with tools/race_probe{2,3}.py: k_vn_gather6_lds (keypoint stage, side stream of BufferPipeline.register_batches) returned wrong
values in 16 lanes of a wavefront in 40-65 % of its launches while a kernel full of v_mfma_f32_16x16x32_f16 (k_nn1f_sweep,
k_cyl_net_h3, k_cost_net_h3) ran on the other stream -- never alone, never beside the fp32-MFMA kernels.  The compiler had built its
inner loop on packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32); with those off (buffer_amd/build.py) the difference is gone.
These tests keep it gone: bitwise-equal outputs of the keypoint-stage kernels beside every chip-filling kernel of the main stream, and
the two-stream software pipeline against the plain batch."""
from dataclasses import replace

import numpy as np
import pytest
import torch

from buffer_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def stage(dev):
    from buffer_amd import ops, pyramid, registration
    from buffer_amd.config import THREEDMATCH
    from buffer_amd.patch_embedder import PatchEmbedder
    from buffer_amd.pipeline import BufferPipeline
    cfg = replace(THREEDMATCH, num_keypts=600)
    pipe = BufferPipeline(cfg, dev)
    pipe.calibrate([synth.make_pair(1000)])
    inps = [pipe.upload(synth.make_pair(2000 + i)) for i in range(4)]
    lens = np.concatenate([np.asarray(i['lengths'], np.int32) for i in inps])
    pts = torch.cat([i['points'] for i in inps])
    feats = torch.cat([i['features'] for i in inps]).contiguous()
    pyr = pyramid.build_pyramid(pts, lens, pipe.limits, cfg)
    PL, P, N, PO = pipe.point, pyr['points'], pyr['neighbors'], pyr['pools']
    x0 = ops.vn_gather_block(PL.b0, P[0], P[0], feats, N[0], 6, PL.scale)
    victims = {
        'k_vn_gather6_lds': lambda: ops.vn_gather_block(PL.b0, P[0], P[0], feats, N[0], 6, PL.scale),
        'k_vn_gather_pre': lambda: ops.vn_gather_block(PL.res[0]['conv'], P[1], P[0], x0, PO[0], 1, PL.scale),
        'k_gather_max': lambda: ops.gather_max(x0, PO[0]),
        'k_vn_pointwise': lambda: ops.vn_pointwise(PL.res[0]['short'], x0),
        'point learner (efcnn + detnet)': lambda: (lambda a: torch.cat([a[0], a[1], pipe.point.detnet(pyr, a[2], a[3])], 1))(pipe.point.efcnn(pyr, feats)),
    }
    g = torch.Generator(device='cpu').manual_seed(0)
    n = 8000
    x = torch.relu(torch.randn((n, 48, 140), generator=g)).to(dev)
    equi = torch.nn.functional.normalize(torch.randn((n, 32, 7, 20), generator=g), dim=1).to(dev)
    srow, trow = torch.randint(0, n, (n,), generator=g).to(dev), torch.randint(0, n, (n,), generator=g).to(dev)
    desc = torch.nn.functional.normalize(torch.randn((2, n, 32), generator=g), dim=2).to(dev)
    pe_f, pe_s = PatchEmbedder(pipe.W, dev, cfg), PatchEmbedder(pipe.W, dev, replace(cfg, cnn_arith='split'))
    cv_f, cv_s = registration.CostVolume(pipe.W, dev, 20, 'f32'), registration.CostVolume(pipe.W, dev, 20, 'split')
    aggressors = {
        'k_nn1f_sweep': lambda: ops.knn(desc[:1], desc[1:], 1),
        'k_cyl_net_h3 + head': lambda: pe_s.fused.with_head(x, pe_s.fused_head),
        'k_cost_net_h3': lambda: cv_s.fused.gathered(equi, srow, trow),
        'k_cyl_net_wg': lambda: pe_f.fused(x),
        'k_cost_net': lambda: cv_f.fused.gathered(equi, srow, trow),
    }
    return pipe, inps, victims, aggressors


@pytest.mark.parametrize('aggressor', ['k_nn1f_sweep', 'k_cyl_net_h3 + head', 'k_cost_net_h3', 'k_cyl_net_wg', 'k_cost_net'])
def test_keypoint_stage_kernels_are_bit_stable_beside_the_matrix_kernels(stage, dev, aggressor):
    pipe, inps, victims, aggressors = stage
    ref = {k: f().clone() for k, f in victims.items()}
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev, priority=-1)
    bad = {k: 0 for k in victims}
    runs = 40
    for _ in range(runs):
        aggressors[aggressor]()                              # a few ms of chip-filling work on the current stream ...
        with torch.cuda.stream(side):                        # ... and the keypoint-stage kernels beside it
            outs = {k: f() for k, f in victims.items()}
        torch.cuda.synchronize()
        for k, o in outs.items():
            bad[k] += 0 if torch.equal(o, ref[k]) else 1
    print(f'CONCURRENCY beside {aggressor}: launches of {runs} that differ from the lone run: {bad}')
    assert not any(bad.values()), bad


@pytest.mark.parametrize('arith', ['f32', 'split'])
def test_two_stream_pipeline_equals_the_plain_batch_bitwise(stage, dev, arith):
    """register_batches (keypoint stage of step i+1 beside the CNN / matching kernels of step i) == register_batch, step by step,
    15 x 3 steps (the split path differed in 5-10 % of such runs before packed fp32 instructions were turned off)."""
    from buffer_amd.pipeline import BufferPipeline
    pipe0, inps, _, _ = stage
    pipe = BufferPipeline(replace(pipe0.cfg, cnn_arith=arith), dev, limits=pipe0.limits)
    ref = torch.stack(pipe.register_batch(inps, seeds=[0, 1, 2, 3]))
    bad = 0
    for _ in range(15):
        out = pipe.register_batches([inps, inps, inps], seeds=[[0, 1, 2, 3]] * 3)
        bad += sum(0 if torch.equal(torch.stack(o), ref) else 1 for o in out)
    assert bad == 0, f'{bad} of 45 pipelined steps differ from the plain batch'
