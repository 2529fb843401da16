"""Evaluation post-processing: the per-sample oracle (oracle/pose_oracle.py) against golden vectors
from the reference's own metrics.py / utils.py / computePositionSPN.py, and the batched product code
(speedplusbaseline_amd/pose.py) against the oracle.  EPnP: OpenCV is third-party and absent (parity unpinned for its
values) -- the product's EPnP is held to exact-pose recovery, to a Levenberg-Marquardt reprojection minimiser on noisy
observations, and to permutation / batch invariance."""
import os

import numpy as np
import pytest

from oracle import pose_oracle as P
from oracle import portable_rng as prng
from speedplusbaseline_amd import pose

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_golden.npz"))
PTS, K, DIST = P.synth_model()


def same_rotation(qa, qb, tol):
    qa, qb = np.atleast_2d(qa), np.atleast_2d(qb)
    d = np.abs(np.sum(qa * qb, axis=1))
    assert np.all(np.abs(d - 1) < tol), np.abs(d - 1).max()


def test_oracle_matches_reference_golden():
    n = len(G["q_gt"])
    for i in range(n):
        assert abs(P.error_orientation(G["q_pr"][i], G["q_gt"][i]) - G["err_q"][i]) < 1e-9
        assert abs(P.error_translation(G["t_pr"][i], G["t_gt"][i]) - G["err_t"][i]) < 1e-12
        assert np.allclose(P.quat2dcm(G["q_gt"][i]), G["dcm"][i], atol=1e-14)
        assert np.allclose(P.project_keypoints(G["q_gt"][i], G["t_gt"][i], K, DIST, PTS), G["proj"][i], atol=1e-9)
        same_rotation(P.weighted_mean_quaternion(G["wm_qs"][i], G["wm_w"][i]), G["wm_q"][i], 1e-12)
        assert np.allclose(P.compute_position_spn(G["wm_q"][i], G["bbox"][i], PTS, K, DIST), G["t_spn"][i], atol=1e-9)
        if G["speed_thr_ran"][i]:       # where the shipped speed_score runs at all (F9), the fixed one agrees with it
            s, a = P.speed_score(G["t_pr"][i], G["q_pr"][i], G["t_gt"][i], G["q_gt"][i], True, 0.169, 0.002173)
            assert abs(s - G["speed_thr"][i]) < 1e-12 and a == G["speed_thr_acc"][i]
    assert bool(G["speed_raw_raises"])      # the reference raises UnboundLocalError for applyThresh=False (metrics.py:62)
    assert 0 < int(G["speed_thr_ran"].sum()) < n


def test_batched_metrics_and_geometry_match_oracle():
    q_gt, t_gt, q_pr, t_pr = G["q_gt"], G["t_gt"], G["q_pr"], G["t_pr"]
    assert np.allclose(pose.error_orientation(q_pr, q_gt), G["err_q"], atol=1e-9)
    assert np.allclose(pose.error_translation(t_pr, t_gt), G["err_t"], atol=1e-12)
    for thr in (False, True):
        s, a = pose.speed_score(t_pr, q_pr, t_gt, q_gt, applyThresh=thr, rotThresh=0.169, posThresh=0.002173)
        ref = [P.speed_score(t_pr[i], q_pr[i], t_gt[i], q_gt[i], thr, 0.169, 0.002173) for i in range(len(q_gt))]
        assert np.allclose(s, [r[0] for r in ref], atol=1e-12) and np.array_equal(a, [r[1] for r in ref])
    assert np.allclose(pose.quat2dcm(q_gt), G["dcm"], atol=1e-14)
    assert np.allclose(pose.project_keypoints(q_gt, t_gt, K, DIST, PTS), G["proj"], atol=1e-9)
    assert np.allclose(pose.project_keypoints(q_gt, t_gt, K, None, PTS.T), G["proj_nodist"], atol=1e-9)
    same_rotation(pose.weighted_mean_quaternion(G["wm_qs"], G["wm_w"]), G["wm_q"], 1e-12)
    same_rotation(pose.weighted_mean_quaternion(G["wm_qs"]), G["wm_q_unweighted"], 1e-12)
    same_rotation(pose.weighted_mean_quaternion(G["wm_qs"][3].T), G["wm_q_unweighted"][3], 1e-12)       # the (4,N) form
    assert np.allclose(pose.compute_position_spn(G["wm_q"], G["bbox"], PTS, K, DIST), G["t_spn"], atol=1e-8)
    assert np.allclose(pose.compute_position_spn(q_gt, G["bbox"], PTS.T, K, DIST), G["t_spn_gtq"], atol=1e-8)
    assert np.allclose(pose.compute_position_spn(q_gt[5], G["bbox"][5], PTS, K, DIST)[0], G["t_spn_gtq"][5], atol=1e-8)   # one sample


def test_spn_attitude_topk_softmax():
    C, k = 300, 5
    qc = prng.uniform("pose/qclass", (C, 4), -1, 1).astype(np.float64); qc /= np.linalg.norm(qc, axis=1, keepdims=True)
    w = prng.uniform("pose/logits", (7, C), -4, 4).astype(np.float64)
    q, top, tw = pose.spn_attitude(w, qc, k)
    for b in range(7):
        qo, topo, two = P.spn_attitude(w[b], qc, k)
        assert np.array_equal(top[b], topo) and np.allclose(tw[b], two, atol=1e-15)
        same_rotation(q[b], qo, 1e-12)
    import torch                                            # the reference takes topk + softmax in torch (inference.py:174-175)
    tv, ti = torch.topk(torch.from_numpy(w), k, dim=1)
    assert np.array_equal(ti.numpy(), top) and np.allclose(torch.softmax(tv, 1).numpy(), tw, atol=1e-15)


def test_rotmat_quat_roundtrip():
    q, _ = P.synth_poses(64, seed=9)
    R = np.transpose(pose.quat2dcm(q), (0, 2, 1))           # camera <- model
    same_rotation(pose.rotmat_to_quat(R), q, 1e-12)
    from scipy.spatial.transform import Rotation
    same_rotation(pose.rotmat_to_quat(R), Rotation.from_matrix(R).as_quat()[:, [3, 0, 1, 2]], 1e-12)


def test_epnp_recovers_exact_poses():
    """noise-free observations through the full camera model (with lens distortion): the pose comes back to 1e-6"""
    q, t = P.synth_poses(64, seed=11)
    px = np.transpose(pose.project_keypoints(q, t, K, DIST, PTS), (0, 2, 1))
    qe, te = pose.epnp(PTS, px, K, DIST)
    assert pose.error_orientation(qe, q).max() < 1e-3                     # degrees (5 undistortion iterations, as OpenCV)
    assert (pose.error_translation(te, t) / np.linalg.norm(t, axis=1)).max() < 2e-5
    px0 = np.transpose(pose.project_keypoints(q, t, K, None, PTS), (0, 2, 1))
    q0, t0 = pose.epnp(PTS, px0, K, None)                                  # pinhole: limited by the arithmetic only
    assert pose.error_orientation(q0, q).max() < 1e-5 and pose.error_translation(t0, t).max() < 1e-7
    q1, t1 = pose.pnp(PTS, px0[7], K)                                      # the reference's one-image signature
    assert np.allclose(q1, q0[7]) and np.allclose(t1, t0[7])
    perm = np.array([3, 0, 10, 5, 1, 9, 2, 8, 4, 7, 6])                    # the point order is immaterial
    q2, t2 = pose.epnp(PTS[perm], px0[:, perm], K, None)
    same_rotation(q2, q0, 1e-9); assert np.allclose(t2, t0, atol=1e-7)
    with pytest.raises(ValueError):
        pose.epnp(PTS[:10], px0, K)


def test_epnp_on_noisy_keypoints_is_close_to_the_reprojection_optimum():
    """sigma = 1 px on a 1920x1200 frame: EPnP lands within a fraction of a degree / a percent of range of the
    Levenberg-Marquardt optimum started from it, and its RMS reprojection error is within 1.5x of the optimum's"""
    q, t = P.synth_poses(32, seed=13)
    px = np.transpose(pose.project_keypoints(q, t, K, None, PTS), (0, 2, 1))
    noise = (prng.uniform("pose/noise", px.shape, -1, 1) + prng.uniform("pose/noise2", px.shape, -1, 1)) * 1.2247   # ~unit variance
    qe, te = pose.epnp(PTS, px + noise, K, None)
    worse = 0
    for i in range(len(q)):
        qo, to, rms_o = P.pnp_refine(PTS, (px + noise)[i], K, qe[i], te[i])
        uv = pose.project_keypoints(qe[i], te[i], K, None, PTS)[0].T
        rms_e = float(np.sqrt(np.mean((uv - (px + noise)[i]) ** 2)))
        assert rms_e < 1.5 * rms_o + 0.2, (i, rms_e, rms_o)
        assert P.error_orientation(qe[i], qo) < 1.0 and P.error_translation(te[i], to) / np.linalg.norm(to) < 0.03
        worse += rms_e > 1.2 * rms_o
    assert worse <= len(q) // 3
    assert pose.error_orientation(qe, q).mean() < 0.6 and (pose.error_translation(te, t) / np.linalg.norm(t, axis=1)).mean() < 0.01


def test_keypoints_to_pixels_matches_oracle():
    x = prng.uniform("pose/kx", (5, 11)); y = prng.uniform("pose/ky", (5, 11))
    bb = np.array([[100, 400, 50, 380], [0, 1920, 0, 1200], [900.5, 1000.25, 600, 650], [10, 20, 30, 45], [5, 1915, 7, 1190]], dtype=np.float64)
    got = pose.keypoints_to_pixels(x, y, bb)
    for b in range(5):
        assert np.allclose(got[b], P.keypoints_to_pixels(x[b], y[b], bb[b]), atol=1e-12)
