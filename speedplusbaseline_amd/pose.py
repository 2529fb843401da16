"""Evaluation post-processing, batched: network outputs -> pose -> SPEED metrics.

Host-side float64 numpy over the whole batch at once -- the reference does this one image at a time on the CPU with
OpenCV / scipy (src/core/inference.py:79-92,170-196,227-249; src/utils/utils.py:139-269; src/utils/metrics.py:30-67;
src/utils/computePositionSPN.py:33-176).  A batch is 11 keypoints (or 5 class quaternions) per image: a few hundred
numbers, which is why this stays on the host next to the result files it feeds; the GPU part of evaluation is the forward.

  * keypoints_to_pixels, epnp                KRN: RoI de-normalisation, then EPnP (Lepetit, Moreno-Noguer, Fua 2009) -- what
                                             the reference obtains from cv2.solvePnP(flags=SOLVEPNP_EPNP): undistort the
                                             observations, four control points from the model's principal axes, the 12x12
                                             null-space problem, three beta initialisations refined by Gauss-Newton, rigid
                                             alignment, best reprojection error wins.  OpenCV is not a dependency here.
  * spn_attitude, weighted_mean_quaternion   SPN: top-k regress logits -> softmax -> eigenvector mean of class quaternions
  * compute_position_spn                     SPN: similar-triangles initial guess + Gauss-Newton on the box residuals
  * error_orientation / error_translation / speed_score   with the reference's F9 defect fixed: metrics.py:53-62 assigns
                                             `speed_r` but sums `speed_q`, so applyThresh=False (and every sample above the
                                             rotation threshold) raises UnboundLocalError there; here speed = speed_t + speed_r.
Quaternions are scalar-first unit quaternions, as everywhere in the reference.
"""
import numpy as np

_PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


# ------------------------------------------------------------------------------------------------------ rotations
def quat2dcm(q):
    """[...,4] -> [...,3,3], the reference's convention (utils.py:169-199: the matrix that maps camera <- body is its
    transpose, which is what project_keypoints uses)"""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    q0, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    d = np.empty(q.shape[:-1] + (3, 3))
    d[..., 0, 0] = 2 * q0 ** 2 - 1 + 2 * q1 ** 2
    d[..., 1, 1] = 2 * q0 ** 2 - 1 + 2 * q2 ** 2
    d[..., 2, 2] = 2 * q0 ** 2 - 1 + 2 * q3 ** 2
    d[..., 0, 1] = 2 * q1 * q2 + 2 * q0 * q3
    d[..., 0, 2] = 2 * q1 * q3 - 2 * q0 * q2
    d[..., 1, 0] = 2 * q1 * q2 - 2 * q0 * q3
    d[..., 1, 2] = 2 * q2 * q3 + 2 * q0 * q1
    d[..., 2, 0] = 2 * q1 * q3 + 2 * q0 * q2
    d[..., 2, 1] = 2 * q2 * q3 - 2 * q0 * q1
    return d


def rotmat_to_quat(Rm):
    """proper rotation matrices [...,3,3] -> scalar-first unit quaternions with w >= 0 (what pnp() returns through
    scipy's Rotation.from_matrix(...).as_quat(), utils.py:262-266, up to the sign convention)"""
    Rm = np.asarray(Rm, dtype=np.float64)
    m00, m11, m22 = Rm[..., 0, 0], Rm[..., 1, 1], Rm[..., 2, 2]
    # the largest of the four candidates keeps the square root well conditioned
    cand = np.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], axis=-1)
    k = np.argmax(cand, axis=-1)
    s = 2.0 * np.sqrt(np.maximum(np.take_along_axis(cand, k[..., None], -1)[..., 0], 1e-300))
    a, b, c = Rm[..., 2, 1] - Rm[..., 1, 2], Rm[..., 0, 2] - Rm[..., 2, 0], Rm[..., 1, 0] - Rm[..., 0, 1]
    d, e, f = Rm[..., 0, 1] + Rm[..., 1, 0], Rm[..., 0, 2] + Rm[..., 2, 0], Rm[..., 1, 2] + Rm[..., 2, 1]
    q = np.empty(Rm.shape[:-2] + (4,))
    for kk, comp in enumerate(((s / 4, a / s, b / s, c / s), (a / s, s / 4, d / s, e / s), (b / s, d / s, s / 4, f / s),
                               (c / s, e / s, f / s, s / 4))):
        sel = k == kk
        for j in range(4):
            q[..., j] = np.where(sel, comp[j], q[..., j]) if kk else comp[j]
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    return np.where(q[..., :1] < 0, -q, q)


def weighted_mean_quaternion(qs, weights=None):
    """mean rotation of [B,N,4] (or [N,4] / [4,N]) unit quaternions: the dominant eigenvector of sum_i w_i q_i q_i^T
    (what scipy's Rotation.mean computes for utils.py:139-166).  Returned with w >= 0; q and -q are the same rotation."""
    qs = np.asarray(qs, dtype=np.float64)
    single = qs.ndim == 2
    if single:
        qs = (qs if qs.shape[1] == 4 else qs.T)[None]
    w = np.ones(qs.shape[:2]) if weights is None else np.asarray(weights, dtype=np.float64).reshape(qs.shape[:2])
    qs = qs / np.linalg.norm(qs, axis=-1, keepdims=True)
    A = np.einsum("bn,bni,bnj->bij", w, qs, qs)
    _, vec = np.linalg.eigh(A)
    q = vec[..., -1]
    q = np.where(q[..., :1] < 0, -q, q)
    return q[0] if single else q


# ------------------------------------------------------------------------------------------------------ camera
def _distort(x0, y0, dist):
    r2 = x0 * x0 + y0 * y0
    cd = 1 + dist[0] * r2 + dist[1] * r2 * r2 + dist[4] * r2 * r2 * r2
    x = x0 * cd + dist[2] * 2 * x0 * y0 + dist[3] * (r2 + 2 * x0 * x0)
    y = y0 * cd + dist[2] * (r2 + 2 * y0 * y0) + dist[3] * 2 * x0 * y0
    return x, y


def _dist5(distCoeffs):
    d = np.zeros(5) if distCoeffs is None else np.asarray(distCoeffs, dtype=np.float64).reshape(-1)
    return np.concatenate([d, np.zeros(max(0, 5 - d.size))])[:5]


def project_keypoints(q, r, cameraMatrix, distCoeffs, keypoints):
    """utils.py:201-235 for a batch: q [B,4], r [B,3], keypoints [N,3] or [3,N] -> pixels [B,2,N]"""
    q, r = np.atleast_2d(np.asarray(q, dtype=np.float64)), np.atleast_2d(np.asarray(r, dtype=np.float64))
    P = np.asarray(keypoints, dtype=np.float64)
    P = P if P.shape[0] == 3 else P.T                                  # [3,N]
    xyz = np.einsum("bji,jn->bin", quat2dcm(q), P) + r[:, :, None]      # dcm^T @ P + r
    x, y = _distort(xyz[:, 0] / xyz[:, 2], xyz[:, 1] / xyz[:, 2], _dist5(distCoeffs))
    K = np.asarray(cameraMatrix, dtype=np.float64)
    return np.stack([K[0, 0] * x + K[0, 2], K[1, 1] * y + K[1, 2]], axis=1)


def undistort_points(px, cameraMatrix, distCoeffs, iters=5):
    """pixels [...,2] -> normalised, undistorted image coordinates: the fixed-point inversion OpenCV's undistortPoints runs
    before EPnP (5 iterations by default)"""
    K = np.asarray(cameraMatrix, dtype=np.float64)
    d = _dist5(distCoeffs)
    xd = (np.asarray(px, dtype=np.float64)[..., 0] - K[0, 2]) / K[0, 0]
    yd = (np.asarray(px, dtype=np.float64)[..., 1] - K[1, 2]) / K[1, 1]
    x, y = xd.copy(), yd.copy()
    if np.any(d != 0):
        for _ in range(iters):
            r2 = x * x + y * y
            icd = 1.0 / (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2)
            dx = 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
            dy = d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
            x, y = (xd - dx) * icd, (yd - dy) * icd
    return np.stack([x, y], axis=-1)


def keypoints_to_pixels(x_pr, y_pr, bbox):
    """inference.py:239-244: [B,K] normalised outputs and RoIs [B,4] = (xmin, xmax, ymin, ymax) -> pixels [B,K,2]"""
    x = np.asarray(x_pr, dtype=np.float64); y = np.asarray(y_pr, dtype=np.float64); b = np.asarray(bbox, dtype=np.float64)
    return np.stack([x * (b[:, 1:2] - b[:, 0:1]) + b[:, 0:1], y * (b[:, 3:4] - b[:, 2:3]) + b[:, 2:3]], axis=-1)


# ------------------------------------------------------------------------------------------------------ EPnP
class _Model:
    """what depends on the 3-D model only: control points (centroid + principal axes) and barycentric coordinates"""

    def __init__(self, pts3d):
        pw = np.asarray(pts3d, dtype=np.float64)
        pw = pw if pw.shape[1] == 3 else pw.T
        n = pw.shape[0]
        c0 = pw.mean(0)
        d = pw - c0
        lam, vec = np.linalg.eigh(d.T @ d)                              # ascending; any orthonormal frame works
        cws = np.vstack([c0] + [c0 + np.sqrt(max(lam[i], 0.0) / n) * vec[:, i] for i in (2, 1, 0)])
        CC = (cws[1:] - cws[0]).T
        a123 = np.linalg.solve(CC, d.T).T
        self.pw, self.cws, self.n = pw, cws, n
        self.alphas = np.hstack([1.0 - a123.sum(1, keepdims=True), a123])          # [n,4]
        self.rho = np.array([np.sum((cws[a] - cws[b]) ** 2) for a, b in _PAIRS])


def _gauss_newton(L, rho, betas, iters=5):
    b = betas.copy()
    for _ in range(iters):
        b0, b1, b2, b3 = b[:, 0:1], b[:, 1:2], b[:, 2:3], b[:, 3:4]
        A = np.stack([2 * L[..., 0] * b0 + L[..., 1] * b1 + L[..., 3] * b2 + L[..., 6] * b3,
                      L[..., 1] * b0 + 2 * L[..., 2] * b1 + L[..., 4] * b2 + L[..., 7] * b3,
                      L[..., 3] * b0 + L[..., 4] * b1 + 2 * L[..., 5] * b2 + L[..., 8] * b3,
                      L[..., 6] * b0 + L[..., 7] * b1 + L[..., 8] * b2 + 2 * L[..., 9] * b3], axis=-1)     # [B,6,4]
        f = rho[None] - (L[..., 0] * b0 * b0 + L[..., 1] * b0 * b1 + L[..., 2] * b1 * b1 + L[..., 3] * b0 * b2 + L[..., 4] * b1 * b2 +
                         L[..., 5] * b2 * b2 + L[..., 6] * b0 * b3 + L[..., 7] * b1 * b3 + L[..., 8] * b2 * b3 + L[..., 9] * b3 * b3)
        b = b + _lstsq(A, f)
    return b


def _lstsq(A, y):
    """batched least squares through the pseudo-inverse (6 equations, <= 5 unknowns)"""
    return np.einsum("bij,bj->bi", np.linalg.pinv(A), y)


def _align(model, V, betas, uv):
    """control points in the camera frame from the betas -> model points -> rigid alignment; returns R, t, mean
    reprojection error in normalised units"""
    ccs = np.einsum("bk,bkj->bj", betas, V).reshape(-1, 4, 3)
    pcs = np.einsum("na,bac->bnc", model.alphas, ccs)
    sgn = np.where(pcs[:, 0, 2] < 0, -1.0, 1.0)[:, None, None]
    pcs = pcs * sgn
    pc0, pw0 = pcs.mean(1), model.pw.mean(0)
    H = np.einsum("bni,nj->bij", pcs - pc0[:, None], model.pw - pw0)
    U, _, Vt = np.linalg.svd(H)
    Rm = U @ Vt
    neg = np.linalg.det(Rm) < 0
    if np.any(neg):
        U = U.copy(); U[neg, :, 2] *= -1
        Rm = U @ Vt
    t = pc0 - np.einsum("bij,j->bi", Rm, pw0)
    cam = np.einsum("bij,nj->bni", Rm, model.pw) + t[:, None]
    err = np.sqrt(((cam[..., :2] / cam[..., 2:3] - uv) ** 2).sum(-1)).mean(1)
    return Rm, t, err


def epnp(points_3D, points_2D, cameraMatrix, distCoeffs=None):
    """pose of a known N-point model from its pixel observations, for a batch.
    points_3D [N,3], points_2D [B,N,2] (pixels) -> (q [B,4] scalar-first, camera <- model as the reference's pnp() returns
    it, utils.py:237-269; t [B,3])"""
    model = points_3D if isinstance(points_3D, _Model) else _Model(points_3D)
    px = np.asarray(points_2D, dtype=np.float64)
    if px.ndim == 2:
        px = px[None]
    if px.shape[1] != model.n:
        raise ValueError("points 3D and points 2D must have same number of vertices")
    uv = undistort_points(px, cameraMatrix, distCoeffs)                                   # [B,N,2]
    B, a = uv.shape[0], model.alphas
    M = np.zeros((B, 2 * model.n, 12))
    for j in range(4):
        M[:, 0::2, 3 * j] = a[None, :, j]; M[:, 0::2, 3 * j + 2] = -a[None, :, j] * uv[..., 0]
        M[:, 1::2, 3 * j + 1] = a[None, :, j]; M[:, 1::2, 3 * j + 2] = -a[None, :, j] * uv[..., 1]
    _, vec = np.linalg.eigh(np.einsum("bki,bkj->bij", M, M))                               # ascending eigenvalues
    V = np.transpose(vec[:, :, :4], (0, 2, 1))                                             # [B,4,12]: v0 = smallest
    Vc = V.reshape(B, 4, 4, 3)
    dv = np.stack([Vc[:, :, p] - Vc[:, :, q_] for p, q_ in _PAIRS], axis=2)               # [B,4 vectors,6 pairs,3]
    dot = lambda i, j: np.einsum("bpc,bpc->bp", dv[:, i], dv[:, j])
    L = np.stack([dot(0, 0), 2 * dot(0, 1), dot(1, 1), 2 * dot(0, 2), 2 * dot(1, 2), dot(2, 2), 2 * dot(0, 3), 2 * dot(1, 3),
                  2 * dot(2, 3), dot(3, 3)], axis=-1)                                     # [B,6,10]
    rho = model.rho
    rhs = np.broadcast_to(rho, (B, 6))
    cands = []
    # N = 1..4 null-space dimensions: linearised unknowns (b11, b12, b13, b14)
    x = _lstsq(L[..., [0, 1, 3, 6]], rhs)
    s = np.where(x[:, 0] < 0, -1.0, 1.0)
    b0 = np.sqrt(np.abs(x[:, 0])) + 1e-300
    cands.append(np.stack([b0, s * x[:, 1] / b0, s * x[:, 2] / b0, s * x[:, 3] / b0], axis=1))
    # N = 2: (b11, b12, b22)
    x = _lstsq(L[..., [0, 1, 2]], rhs)
    neg = x[:, 0] < 0
    b0 = np.sqrt(np.abs(x[:, 0]))
    b1 = np.where(neg, np.sqrt(np.maximum(-x[:, 2], 0)), np.sqrt(np.maximum(x[:, 2], 0)))
    b0 = np.where(x[:, 1] < 0, -b0, b0)
    cands.append(np.stack([b0, b1, np.zeros(B), np.zeros(B)], axis=1))
    # N = 3: (b11, b12, b22, b13, b23)
    x = _lstsq(L[..., [0, 1, 2, 3, 4]], rhs)
    neg = x[:, 0] < 0
    b0 = np.sqrt(np.abs(x[:, 0]))
    b1 = np.where(neg, np.sqrt(np.maximum(-x[:, 2], 0)), np.sqrt(np.maximum(x[:, 2], 0)))
    b0 = np.where(x[:, 1] < 0, -b0, b0)
    cands.append(np.stack([b0, b1, x[:, 3] / np.where(b0 == 0, 1e-300, b0), np.zeros(B)], axis=1))
    best = None
    for betas in cands:
        Rm, t, err = _align(model, V, _gauss_newton(L, rho, betas), uv)
        if best is None:
            best = [Rm, t, err]
        else:
            take = err < best[2]
            best[0] = np.where(take[:, None, None], Rm, best[0]); best[1] = np.where(take[:, None], t, best[1])
            best[2] = np.where(take, err, best[2])
    return rotmat_to_quat(best[0]), best[1]


def pnp(points_3D, points_2D, cameraMatrix, distCoeffs=None):
    """the reference's per-sample signature (utils.py:237-269): one image, (q [4], t [3])"""
    q, t = epnp(points_3D, np.asarray(points_2D, dtype=np.float64).reshape(1, -1, 2), cameraMatrix, distCoeffs)
    return q[0], t[0]


# ------------------------------------------------------------------------------------------------------ SPN
def spn_attitude(weights, q_class, k):
    """inference.py:174-181 for a batch: regress logits [B,C] -> top-k -> softmax -> weighted mean of the class quaternions"""
    w = np.asarray(weights, dtype=np.float64)
    top = np.argsort(-w, axis=1, kind="stable")[:, :k]
    tw = np.take_along_axis(w, top, 1)
    tw = np.exp(tw - tw.max(1, keepdims=True)); tw /= tw.sum(1, keepdims=True)
    return weighted_mean_quaternion(np.asarray(q_class, dtype=np.float64)[top], tw), top, tw


def compute_position_spn(q, bbox, corners3D, cameraMatrix, distCoeffs=None, max_model_length=1.246, max_iter=50, tol=5e-10):
    """computePositionSPN.py:33-176 for a batch: q [B,4], bbox [B,4] = (xmin, xmax, ymin, ymax) -> position [B,3].
    Initial guess by similar triangles on the box diagonal, then Gauss-Newton on the four box-edge residuals of the model's
    extremal points (Jacobian without the distortion terms and rounded to float32, as the reference builds it); every
    sample iterates until its own step falls under `tol` or `max_iter` + 1 iterations have run."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float64)); bb = np.atleast_2d(np.asarray(bbox, dtype=np.float64))
    K = np.asarray(cameraMatrix, dtype=np.float64); d = _dist5(distCoeffs)
    P = np.asarray(corners3D, dtype=np.float64); P = P if P.shape[0] == 3 else P.T
    B = q.shape[0]
    w, h = bb[:, 1] - bb[:, 0], bb[:, 3] - bb[:, 2]
    az = np.arctan((bb[:, 0] + w / 2 - K[0, 2]) / K[0, 0]); el = np.arctan((bb[:, 2] + h / 2 - K[1, 2]) / K[1, 1])
    rng = K[0, 0] * max_model_length / np.sqrt(w * w + h * h)
    # Ry(-az) @ Rx(-el) @ [0, 0, range]
    beta = np.stack([-np.sin(az) * np.cos(el), np.sin(el), np.cos(az) * np.cos(el)], axis=1) * rng[:, None]
    Pv = np.einsum("bji,jn->bin", quat2dcm(q), P)                       # model points in the camera frame, [B,3,N]
    active = np.ones(B, dtype=bool)
    ar = np.arange(B)
    for _ in range(max_iter + 1):
        img = project_keypoints(q, beta, K, np.zeros(5), P)              # extremal points: no distortion (as the reference)
        idx = np.stack([img[:, 0].argmin(1), img[:, 0].argmax(1), img[:, 1].argmin(1), img[:, 1].argmax(1)], axis=1)   # left, right, top, bottom
        X = np.stack([Pv[ar, :, idx[:, i]] for i in range(4)], axis=1)   # [B,4,3]
        den = X[..., 2] + beta[:, 2:3]
        x0, y0 = (X[..., 0] + beta[:, 0:1]) / den, (X[..., 1] + beta[:, 1:2]) / den
        xd, yd = _distort(x0, y0, d)
        u, v = K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]
        r = np.stack([u[:, 0] - bb[:, 0], u[:, 1] - bb[:, 1], v[:, 2] - bb[:, 2], v[:, 3] - bb[:, 3]], axis=1)
        J = np.zeros((B, 4, 3))
        J[:, :2, 0] = K[0, 0] / den[:, :2]; J[:, :2, 2] = -K[0, 0] * (X[:, :2, 0] + beta[:, 0:1]) / den[:, :2] ** 2
        J[:, 2:, 1] = K[1, 1] / den[:, 2:]; J[:, 2:, 2] = -K[1, 1] * (X[:, 2:, 1] + beta[:, 1:2]) / den[:, 2:] ** 2
        J = J.astype(np.float32).astype(np.float64)
        JtJ = np.einsum("bki,bkj->bij", J, J)
        step = np.einsum("bij,bj->bi", np.linalg.inv(JtJ), np.einsum("bki,bk->bi", J, r))
        new = np.where(active[:, None], beta - step, beta)
        active = active & (np.linalg.norm(new - beta, axis=1) > tol)
        beta = new
        if not active.any():
            break
    return beta


# ------------------------------------------------------------------------------------------------------ metrics
def error_translation(t_pr, t_gt):
    """metrics.py:30-34 for a batch: [B,3] x [B,3] -> [B] metres"""
    return np.sqrt(np.sum(np.square(np.asarray(t_gt, dtype=np.float64).reshape(-1, 3) - np.asarray(t_pr, dtype=np.float64).reshape(-1, 3)), axis=1))


def error_orientation(q_pr, q_gt):
    """metrics.py:36-43 for a batch: [B] degrees"""
    qd = np.abs(np.sum(np.asarray(q_pr, dtype=np.float64).reshape(-1, 4) * np.asarray(q_gt, dtype=np.float64).reshape(-1, 4), axis=1))
    return np.rad2deg(2 * np.arccos(np.minimum(qd, 1.0)))


def speed_score(t_pr, q_pr, t_gt, q_gt, applyThresh=True, rotThresh=0.5, posThresh=0.005):
    """metrics.py:45-67 for a batch, F9 fixed: (speed [B], acc [B])"""
    err_t, err_q = error_translation(t_pr, t_gt), error_orientation(q_pr, q_gt)
    speed_t = err_t / np.sqrt(np.sum(np.square(np.asarray(t_gt, dtype=np.float64).reshape(-1, 3)), axis=1))
    speed_r = np.deg2rad(err_q)
    if applyThresh:
        speed_r = np.where(err_q < rotThresh, 0.0, speed_r)
        speed_t = np.where(speed_t < posThresh, 0.0, speed_t)
    acc = ((err_q < rotThresh) & (speed_t < posThresh)).astype(np.float64)
    return speed_t + speed_r, acc
