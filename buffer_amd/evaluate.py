"""A-H -- evaluation harness: 3DMatch `.log` trajectory IO and the Registration-Recall protocol
(counterpart of ThreeDMatch/test.py:18-196,242-308).  Pure host code."""
import math
import os

import numpy as np


def read_trajectory(filename, dim=4):
    """test.py:18-55: -> (pairs str[n,3], traj f32[n,dim,dim])."""
    with open(filename) as f:
        lines = f.readlines()
    keys = [[t.strip() for t in ln.split('\t')[0:3]] for ln in lines[0::(dim + 1)]]
    traj = [ln.split('\t')[0:dim] for i, ln in enumerate(lines) if i % (dim + 1) != 0]
    return np.asarray(keys), np.asarray(traj, dtype=np.float32).reshape(-1, dim, dim)


def read_trajectory_info(filename, dim=6):
    """test.py:58-89: -> (n_fragments, info f32[n,6,6])."""
    with open(filename) as f:
        contents = f.readlines()
    n_pairs = len(contents) // 7
    assert len(contents) == 7 * n_pairs
    infos, n_frame = [], 0
    for i in range(n_pairs):
        _, _, n_frame = [int(x) for x in contents[i * 7].strip().split()]
        infos.append(np.concatenate([np.array(ln.split(), dtype=np.float64).reshape(1, -1)
                                     for ln in contents[i * 7 + 1:i * 7 + 7]], axis=0))
    return n_frame, np.asarray(infos, dtype=np.float32).reshape(-1, dim, dim)


def append_log(path, src_id, tgt_id, pose_est):
    """test.py:252-261: the log holds the INVERSE of the estimated pose."""
    trans = np.linalg.inv(np.asarray(pose_est, np.float64))
    os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
    with open(path, 'a+') as f:
        f.write(f'{src_id}\t {tgt_id}\t  1\n')
        for r in range(4):
            f.write(f"{trans[r, 0]}\t {trans[r, 1]}\t {trans[r, 2]}\t {trans[r, 3]}\t \n")


def mat2quat(M):
    """nibabel.quaternions.mat2quat (Bar-Itzhack), w >= 0 -- used by test.py:105."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, np.float64).flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0], [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def transformation_error(trans, info):
    """test.py:92-111."""
    er = np.concatenate([trans[:3, 3], mat2quat(trans[:3, :3])[1:]], axis=0)
    return (er.reshape(1, 6) @ info @ er.reshape(6, 1) / info[0, 0]).item()


def evaluate_registration(num_fragment, result, result_pairs, gt_pairs, gt, gt_info, err2=0.2):
    """test.py:114-173: -> (precision, recall, flags, errors); only non-consecutive pairs count."""
    err2 = err2 ** 2
    gt_mask = np.zeros((num_fragment, num_fragment), dtype=np.int64)
    for idx in range(gt_pairs.shape[0]):
        i, j = int(gt_pairs[idx, 0]), int(gt_pairs[idx, 1])
        if j - i > 1:
            gt_mask[i, j] = idx
    n_gt = np.sum(gt_mask > 0)
    errors = np.full(result_pairs.shape[0], np.nan)
    good, n_res, flags = 0, 0, []
    for idx in range(result_pairs.shape[0]):
        i, j = int(result_pairs[idx, 0]), int(result_pairs[idx, 1])
        if gt_mask[i, j] > 0:
            n_res += 1
            g = gt_mask[i, j]
            p = transformation_error(np.linalg.inv(gt[g]) @ result[idx], gt_info[g])
            errors[idx] = p
            if p <= err2:
                good += 1
                flags.append(0)
            else:
                flags.append(1)
        else:
            flags.append(2)
    if n_res == 0:
        n_res += 1e6
    return good * 1.0 / n_res, good * 1.0 / n_gt, flags, errors


def registration_recall(gt_root, log_root, log_name):
    """test.py:287-308: mean over scenes of the per-scene recall."""
    recalls = []
    for scene in sorted(os.listdir(gt_root)):
        gt_pairs, gt_traj = read_trajectory(os.path.join(gt_root, scene, 'gt.log'))
        n_frag, gt_cov = read_trajectory_info(os.path.join(gt_root, scene, 'gt.info'))
        est_pairs, est_traj = read_trajectory(os.path.join(log_root, scene, log_name))
        recalls.append(evaluate_registration(n_frag, est_traj, est_pairs, gt_pairs, gt_traj, gt_cov)[1])
    return float(np.mean(recalls)), recalls


def dgr_success(pose_est, pose_gt, rte_thresh=0.3, rre_thresh=15.0):
    """test.py:263-270 -> (ok, rte, rre_deg)."""
    pose_est, pose_gt = np.asarray(pose_est, np.float64), np.asarray(pose_gt, np.float64)
    rte = np.linalg.norm(pose_est[:3, 3] - pose_gt[:3, 3])
    rre = np.arccos(np.clip((np.trace(pose_est[:3, :3].T @ pose_gt[:3, :3]) - 1) / 2, -1 + 1e-16, 1 - 1e-16)) * 180 / math.pi
    return bool(rte < rte_thresh and rre < rre_thresh), float(rte), float(rre)
