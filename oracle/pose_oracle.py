"""TEST INFRASTRUCTURE (oracle/): per-sample CPU restatement of the reference's evaluation post-processing.

Only tests/ may import this module -- never the product (speedplusbaseline_amd/pose.py is the batched product code).

What it restates (reference file:line), one sample at a time, in float64 numpy exactly as the reference computes it:
  * error_translation, error_orientation, speed_score                     src/utils/metrics.py:30-67
      (F9: the reference's speed_score reads `speed_q`, which only exists when applyThresh and err_q < rotThresh, so
       applyThresh=False raises UnboundLocalError at metrics.py:62; the evident intent -- speed = speed_t + speed_r with
       speed_r zeroed below the threshold -- is what `speed_score` here returns, and the golden generator records both the
       reference's exception and its values on the branch where it does run)
  * weighted_mean_quaternion, quat2dcm, project_keypoints                  src/utils/utils.py:139-235
  * keypoints -> pixels (_keypts_to_pose's RoI step)                      src/core/inference.py:227-249
  * compute_position_spn (similar-triangles guess + Gauss-Newton)         src/utils/computePositionSPN.py:33-176
  * the SPN attitude post-processing (top-k -> softmax -> class quaternions) src/core/inference.py:170-181
Pinned by tests/golden/pose_golden.npz (tests/golden/make_golden_pose.py imports the reference's own functions; cv2 is
stubbed there because only `pnp` touches it).  `pnp` itself is OpenCV's solvePnP(SOLVEPNP_EPNP): third-party, OpenCV
4.5.1.48 per requirements.txt:2, not installed here -> PARITY UNPINNED for the EPnP values; the product's EPnP is held
to exact-pose recovery and to `pnp_refine` below (a Levenberg-Marquardt reprojection minimiser) on noisy keypoints.
"""
import numpy as np
from scipy.spatial.transform import Rotation as R


def error_translation(t_pr, t_gt):
    t_pr = np.reshape(t_pr, (3,)); t_gt = np.reshape(t_gt, (3,))
    return np.sqrt(np.sum(np.square(t_gt - t_pr)))


def error_orientation(q_pr, q_gt):
    q_pr = np.reshape(q_pr, (4,)); q_gt = np.reshape(q_gt, (4,))
    qdot = np.minimum(np.abs(np.dot(q_pr, q_gt)), 1.0)
    return np.rad2deg(2 * np.arccos(qdot))


def speed_score(t_pr, q_pr, t_gt, q_gt, applyThresh=True, rotThresh=0.5, posThresh=0.005):
    err_t = error_translation(t_pr, t_gt)
    err_q = error_orientation(q_pr, q_gt)
    t_gt = np.reshape(t_gt, (3,))
    speed_t = err_t / np.sqrt(np.sum(np.square(t_gt)))
    speed_r = np.deg2rad(err_q)
    if applyThresh and err_q < rotThresh:
        speed_r = 0.0
    if applyThresh and speed_t < posThresh:
        speed_t = 0.0
    speed = speed_t + speed_r
    acc = float(err_q < rotThresh and speed_t < posThresh)
    return speed, acc


def weighted_mean_quaternion(qs, weights=None):
    qs = np.asarray(qs, dtype=np.float64)
    if qs.shape[1] != 4:
        qs = qs.T
    qs = qs[:, [1, 2, 3, 0]]
    if weights is None:
        weights = np.ones((qs.shape[0],))
    q = R.from_quat(qs).mean(weights).as_quat()
    return q[[3, 0, 1, 2]]


def quat2dcm(q):
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q)
    q0, q1, q2, q3 = q
    d = np.zeros((3, 3))
    d[0, 0] = 2 * q0 ** 2 - 1 + 2 * q1 ** 2
    d[1, 1] = 2 * q0 ** 2 - 1 + 2 * q2 ** 2
    d[2, 2] = 2 * q0 ** 2 - 1 + 2 * q3 ** 2
    d[0, 1] = 2 * q1 * q2 + 2 * q0 * q3
    d[0, 2] = 2 * q1 * q3 - 2 * q0 * q2
    d[1, 0] = 2 * q1 * q2 - 2 * q0 * q3
    d[1, 2] = 2 * q2 * q3 + 2 * q0 * q1
    d[2, 0] = 2 * q1 * q3 + 2 * q0 * q2
    d[2, 1] = 2 * q2 * q3 - 2 * q0 * q1
    return d


def project_keypoints(q, r, cameraMatrix, distCoeffs, keypoints):
    keypoints = np.asarray(keypoints, dtype=np.float64)
    if keypoints.shape[0] != 3:
        keypoints = keypoints.T
    kp = np.vstack((keypoints, np.ones((1, keypoints.shape[1]))))
    pose = np.hstack((quat2dcm(q).T, np.expand_dims(np.asarray(r, dtype=np.float64), 1)))
    xyz = pose @ kp
    x0, y0 = xyz[0] / xyz[2], xyz[1] / xyz[2]
    r2 = x0 * x0 + y0 * y0
    cd = 1 + distCoeffs[0] * r2 + distCoeffs[1] * r2 * r2 + distCoeffs[4] * r2 * r2 * r2
    x = x0 * cd + distCoeffs[2] * 2 * x0 * y0 + distCoeffs[3] * (r2 + 2 * x0 * x0)
    y = y0 * cd + distCoeffs[2] * (r2 + 2 * y0 * y0) + distCoeffs[3] * 2 * x0 * y0
    return np.vstack((cameraMatrix[0, 0] * x + cameraMatrix[0, 2], cameraMatrix[1, 1] * y + cameraMatrix[1, 2]))


def keypoints_to_pixels(x_pr, y_pr, bbox):
    """inference.py:239-244: normalised network outputs -> pixels of the full frame through the RoI [xmin, xmax, ymin, ymax]"""
    xmin, xmax, ymin, ymax = [float(v) for v in bbox]
    c = np.stack([np.asarray(x_pr, dtype=np.float64), np.asarray(y_pr, dtype=np.float64)], axis=1)
    c[:, 0] = c[:, 0] * (xmax - xmin) + xmin
    c[:, 1] = c[:, 1] * (ymax - ymin) + ymin
    return c


def spn_attitude(weights_row, q_class, k):
    """inference.py:174-181 for one sample: top-k regress logits -> softmax -> weighted mean of the class quaternions"""
    w = np.asarray(weights_row, dtype=np.float64)
    top = np.argsort(-w, kind="stable")[:k]
    tw = np.exp(w[top] - w[top].max()); tw = tw / tw.sum()
    return weighted_mean_quaternion(np.asarray(q_class)[top], tw), top, tw


# ---- computePositionSPN.py:33-176
def _extremal(q, beta, pts, K):
    img = project_keypoints(q, beta, K, np.zeros(5), pts)
    i1, i2, i3, i4 = np.argmin(img[0]), np.argmin(img[1]), np.argmax(img[0]), np.argmax(img[1])
    p = np.asarray(pts, dtype=np.float64)
    if p.shape[0] != 3:
        p = p.T
    pv = quat2dcm(q).T @ p
    return np.stack([pv[:, i1], pv[:, i3], pv[:, i2], pv[:, i4]])


def _residuals(X, K, dist, beta, bbox):
    Tx, Ty, Tz = beta
    xs, ys = [], []
    for i in range(4):
        Rx, Ry, Rz = X[i]
        x0 = (Rx + Tx) / (Rz + Tz); y0 = (Ry + Ty) / (Rz + Tz)
        r2 = x0 * x0 + y0 * y0
        cd = 1 + dist[0] * r2 + dist[1] * r2 * r2 + dist[4] * r2 * r2 * r2
        x = x0 * cd + dist[2] * 2 * x0 * y0 + dist[3] * (r2 + 2 * x0 * x0)
        y = y0 * cd + dist[2] * (r2 + 2 * y0 * y0) + dist[3] * 2 * x0 * y0
        xs.append(K[0, 0] * x + K[0, 2]); ys.append(K[1, 1] * y + K[1, 2])
    return np.array([xs[0] - bbox[0], xs[1] - bbox[1], ys[2] - bbox[2], ys[3] - bbox[3]])


def _jacobian(X, K, beta):
    fx, fy = K[0, 0], K[1, 1]
    Tx, Ty, Tz = beta
    J = np.array([[fx / (X[0, 2] + Tz), 0, -fx * (X[0, 0] + Tx) / (X[0, 2] + Tz) ** 2],
                  [fx / (X[1, 2] + Tz), 0, -fx * (X[1, 0] + Tx) / (X[1, 2] + Tz) ** 2],
                  [0, fy / (X[2, 2] + Tz), -fy * (X[2, 1] + Ty) / (X[2, 2] + Tz) ** 2],
                  [0, fy / (X[3, 2] + Tz), -fy * (X[3, 1] + Ty) / (X[3, 2] + Tz) ** 2]], dtype=np.float32)   # float32 as the reference
    return J


def compute_position_spn(q, bbox, pts, K, dist=np.zeros(5), max_model_length=1.246):
    dist = np.asarray(dist, dtype=np.float64).reshape(-1)
    xmin, ymin, width, height = bbox[0], bbox[2], bbox[1] - bbox[0], bbox[3] - bbox[2]
    size = np.sqrt(width ** 2 + height ** 2)
    cx, cy = xmin + width / 2.0, ymin + height / 2.0
    az = np.arctan((cx - K[0, 2]) / K[0, 0]); el = np.arctan((cy - K[1, 2]) / K[1, 1])
    rng = K[0, 0] * max_model_length / size
    beta_old = np.squeeze(R.from_euler('y', -az).as_matrix() @ R.from_euler('x', -el).as_matrix() @ np.reshape(np.array([0, 0, rng]), (3, 1)))
    it, dx = 0, 1 + 1e-15
    beta_new = beta_old
    while dx > 5e-10 and it <= 50:
        X = _extremal(q, beta_old, pts, K)
        r = _residuals(X, K, dist, beta_old, bbox)
        J = _jacobian(X, K, beta_old)
        beta_new = beta_old - np.squeeze(np.linalg.inv(J.T @ J) @ J.T @ np.reshape(r, (4, 1)))
        dx = np.linalg.norm(beta_new - beta_old)
        it += 1
        beta_old = beta_new
    return beta_new


# ---- yardstick for the EPnP product code (not a restatement of OpenCV): reprojection-error minimiser from a given start
def pnp_refine(pts3d, pts2d, K, q0, t0):
    from scipy.optimize import least_squares
    pts3d = np.asarray(pts3d, dtype=np.float64)

    def res(p):
        rot = R.from_rotvec(p[:3]).as_matrix()
        xyz = pts3d @ rot.T + p[3:]
        u = K[0, 0] * xyz[:, 0] / xyz[:, 2] + K[0, 2]; v = K[1, 1] * xyz[:, 1] / xyz[:, 2] + K[1, 2]
        return np.concatenate([u - pts2d[:, 0], v - pts2d[:, 1]])
    rv0 = R.from_quat(np.asarray(q0)[[1, 2, 3, 0]]).as_rotvec()
    sol = least_squares(res, np.concatenate([rv0, np.asarray(t0, dtype=np.float64)]), method="lm", xtol=1e-14, ftol=1e-14)
    q = R.from_rotvec(sol.x[:3]).as_quat()[[3, 0, 1, 2]]
    return q, sol.x[3:], float(np.sqrt(np.mean(sol.fun ** 2)))


# ---- synthetic stand-ins for the data assets (tangoPoints.mat, camera.json, attitudeClasses.mat are not redistributed)
def synth_model(seed=3):
    """11 non-coplanar 3-D keypoints of a ~1.2 m object, a 1920x1200 pinhole camera with mild distortion"""
    from . import portable_rng as prng
    pts = prng.uniform("pose/pts", (11, 3), -0.6, 0.6, seed).astype(np.float64)
    K = np.array([[2988.58, 0.0, 960.0], [0.0, 2988.34, 600.0], [0.0, 0.0, 1.0]])
    dist = np.array([-0.2238, 0.5141, -0.0006, -0.0002, -0.1313])
    return pts, K, dist


def synth_poses(n, seed=5):
    """unit quaternions (scalar first) and positions in front of the camera (4..12 m, inside the field of view)"""
    from . import portable_rng as prng
    q = prng.uniform("pose/q", (n, 4), -1.0, 1.0, seed).astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = np.stack([prng.uniform("pose/tx", (n,), -0.6, 0.6, seed), prng.uniform("pose/ty", (n,), -0.4, 0.4, seed),
                  prng.uniform("pose/tz", (n,), 4.0, 12.0, seed)], axis=1).astype(np.float64)
    return q, t
