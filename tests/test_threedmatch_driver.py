"""The 3DMatch test-set driver (file layout, PLY IO, .log writer, RR evaluator) end to end on a synthetic mini
dataset in the reference's directory layout (ThreeDMatch/dataset.py:47-76, test.py:199-308)."""
import os

import numpy as np
import pytest


def test_ply_roundtrip_all_formats(tmp_path):
    from buffer_amd import threedmatch as tdm
    pts = np.random.default_rng(0).normal(size=(257, 3)).astype(np.float32)
    tdm.write_ply(str(tmp_path / 'a.ply'), pts)
    np.testing.assert_array_equal(tdm.read_ply(str(tmp_path / 'a.ply')), pts)
    # ascii with extra properties and comments; big-endian doubles with colours
    with open(tmp_path / 'b.ply', 'w') as f:
        f.write('ply\nformat ascii 1.0\ncomment made by a test\nelement vertex 3\nproperty float x\nproperty float y\n'
                'property float z\nproperty uchar red\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n')
        for r in pts[:3]:
            f.write(f'{r[0]!r} {r[1]!r} {r[2]!r} 7\n'.replace('np.float32(', '').replace(')', ''))
    np.testing.assert_allclose(tdm.read_ply(str(tmp_path / 'b.ply')), pts[:3], rtol=1e-6)
    dt = np.dtype([('nx', '>f4'), ('x', '>f8'), ('y', '>f8'), ('z', '>f8'), ('red', 'u1')])
    rows = np.zeros(5, dt)
    rows['x'], rows['y'], rows['z'] = pts[:5, 0], pts[:5, 1], pts[:5, 2]
    with open(tmp_path / 'c.ply', 'wb') as f:
        f.write(b'ply\nformat binary_big_endian 1.0\nelement vertex 5\nproperty float nx\nproperty double x\n'
                b'property double y\nproperty double z\nproperty uchar red\nend_header\n')
        f.write(rows.tobytes())
    np.testing.assert_array_equal(tdm.read_ply(str(tmp_path / 'c.ply')), pts[:5])
    with open(tmp_path / 'd.ply', 'wb') as f:
        f.write(b'plx\n')
    with pytest.raises(ValueError):
        tdm.read_ply(str(tmp_path / 'd.ply'))


def _mini_dataset(root, scenes, seed):
    """Three overlapping views of a synthetic room per scene, stored like the 3DMatch test split."""
    from buffer_amd import synth, threedmatch as tdm
    rng = np.random.default_rng(seed)
    for scene in scenes:
        size = (2.4, 1.9, 1.7)
        rects = synth.make_scene(rng, size, 6)
        width = 1.5
        poses = []
        for k, lo in enumerate((0.0, 0.45, 0.9)):
            pts, _ = synth.sample_scene(rng, rects, 260_000)
            pts = pts[(pts[:, 0] >= lo) & (pts[:, 0] <= lo + width)]
            sensor = np.array([lo + 0.5 * width, 0.55 * size[1], 0.5 * size[2]])
            R = synth.random_rotation(rng, 0.6)
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = R, -R @ sensor                   # world -> fragment frame
            poses.append(T)
            tdm.write_ply(os.path.join(root, 'test', '3DMatch', 'fragments', scene, f'cloud_bin_{k}.ply'), pts @ R.T + T[:3, 3])
        gtdir = os.path.join(root, 'test', '3DMatch', 'gt_result', scene)
        os.makedirs(gtdir, exist_ok=True)
        with open(os.path.join(gtdir, 'gt.log'), 'w') as fl, open(os.path.join(gtdir, 'gt.info'), 'w') as fi:
            for i, j in ((0, 1), (0, 2), (1, 2)):
                Tij = poses[i] @ np.linalg.inv(poses[j])          # fragment j -> fragment i (3DMatch gt.log convention)
                fl.write(f'{i}\t {j}\t  3\n')
                for r in range(4):
                    fl.write('\t '.join(repr(float(x)) for x in Tij[r]) + '\t \n')
                fi.write(f'{i}\t{j}\t3\n')
                for r in range(6):
                    fi.write('\t'.join('1.0' if c == r else '0.0' for c in range(6)) + '\n')


@pytest.mark.gpu
def test_threedmatch_cli(tmp_path, dev, capsys):
    """python -m buffer_amd.threedmatch on the mini dataset: one JSON line with RR and the limits it calibrated."""
    import json
    from buffer_amd import threedmatch as tdm
    root = str(tmp_path / 'data')
    _mini_dataset(root, tdm.SCENES, seed=5)
    tdm.main(['--root', root, '--log-root', str(tmp_path / 'logs'), '--log-name', 'cli.log', '--batch', '8'])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out['pairs'] == 24 and out['registration_recall'] >= 0.85 and len(out['limits']) == 3, out


@pytest.mark.gpu
def test_threedmatch_layout_end_to_end(tmp_path, dev):
    from buffer_amd import evaluate, threedmatch as tdm
    from buffer_amd.pipeline import BufferPipeline
    root = str(tmp_path / 'data')
    scenes = ['7-scenes-redkitchen', 'sun3d-hotel_uc-scan3']
    _mini_dataset(root, scenes, seed=3)
    ds = tdm.ThreeDMatchTestSet(root, scenes=scenes)
    assert len(ds) == 6 and ds.files[1][0].endswith('cloud_bin_0') and ds.files[1][1].endswith('cloud_bin_2')
    pipe = BufferPipeline(device=dev)
    first = ds.item(0, dev)
    host = {k: (v.cpu().numpy() if hasattr(v, 'cpu') else v) for k, v in first.items()}
    pipe.calibrate([host])
    log_root = str(tmp_path / 'log_3DMatch')
    poses = tdm.register_pairs(pipe, ds, range(len(ds)), batch=4).cpu().numpy()
    stats = tdm.write_logs(ds, poses, log_root, 'run.log')
    out = tdm.summarize(ds, stats, log_root, 'run.log')
    assert out['pairs'] == 6 and out['dgr_recall'] >= 5 / 6, out
    assert out['registration_recall'] == 1.0, out              # the non-consecutive pair (0,2) of both scenes
    # the log holds the inverse of the estimate, in file order (test.py:247-261)
    keys, traj = evaluate.read_trajectory(os.path.join(log_root, scenes[0], 'run.log'))
    assert [tuple(k[:2]) for k in keys] == [('0', '1'), ('0', '2'), ('1', '2')]
    np.testing.assert_allclose(np.linalg.inv(traj[1].astype(np.float64)), poses[1], rtol=0, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_threedmatch_cli_gloo_ranks_write_the_same_logs(tmp_path, dev, world):
    """python -m buffer_amd.threedmatch under 2 and under 8 ranks (sharing the one device, BUFFER_DIST_BACKEND=gloo): pair i ->
    rank i mod W, one all_gather of poses, rank 0 writes the logs -- same files as the 1-rank run (poses to fp32 round-off: the
    stacked launches of the two runs group the pairs differently)."""
    import json
    import subprocess
    import sys
    from buffer_amd import evaluate, threedmatch as tdm
    root = str(tmp_path / 'data')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _mini_dataset(root, tdm.SCENES, seed=7)
    common = ['--root', root, '--log-name', 'run.log', '--batch', '4', '--limits', '17,20,24']
    env = dict(os.environ, BUFFER_DIST_BACKEND='gloo', PYTHONPATH=repo)
    one = subprocess.run([sys.executable, '-m', 'buffer_amd.threedmatch', '--log-root', str(tmp_path / 'one')] + common,
                         capture_output=True, text=True, env=env, cwd=repo, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
                          '--master-port', str(29547 + world), '-m', 'buffer_amd.threedmatch', '--log-root', str(tmp_path / 'two')] + common,
                         capture_output=True, text=True, env=env, cwd=repo, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    o1 = json.loads([l for l in one.stdout.splitlines() if l.startswith('{')][-1])
    o2 = json.loads([l for l in two.stdout.splitlines() if l.startswith('{')][-1])
    assert o1['pairs'] == o2['pairs'] == 24 and o2['n_gpus'] == world and o1['n_gpus'] == 1
    assert o1['registration_recall'] == o2['registration_recall'] and o1['dgr_recall'] == o2['dgr_recall']
    worst = 0.0
    for scene in tdm.SCENES:
        k1, t1 = evaluate.read_trajectory(os.path.join(str(tmp_path / 'one'), scene, 'run.log'))
        k2, t2 = evaluate.read_trajectory(os.path.join(str(tmp_path / 'two'), scene, 'run.log'))
        assert [tuple(k) for k in k1] == [tuple(k) for k in k2]
        worst = max(worst, float(np.abs(np.asarray(t1, np.float64) - np.asarray(t2, np.float64)).max()))
    print(f'1-rank vs {world}-rank log difference:', worst)
    assert worst < 1e-4
