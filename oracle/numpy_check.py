"""Independent NumPy/SciPy re-derivation of one GN iteration of the reference's CT-ICP
(src/ct_icp/ct_icp.cpp:745-981) — TEST INFRASTRUCTURE, round 1's anchor of oracle/ctgn_oracle.c (the reference ships no
golden vectors for this path; since round 2 the oracle is also held to the reference's own sources, oracle/_ref).

It deliberately shares no code with the C oracle or with ct_icp_amd: brute-force neighbour search over the
exported map points, numpy.linalg.svd for the normal (the reference uses Eigen::JacobiSVD), numpy.linalg.solve
for the 12x12 system, scipy Rotation for the SE(3) algebra. tests/golden/make_golden.py stores its outputs as the
committed golden vectors.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation, Slerp


def voxel_coords(p, size):
    return np.trunc(np.asarray(p, float) / size).astype(np.int64)       # int(p / size): truncation toward zero


def alpha(t, tb, te):
    t = np.asarray(t, float)
    lo, hi = min(tb, te), max(tb, te)
    a = np.ones_like(t) if lo == hi else (t - lo) / (hi - lo)
    return np.where((t < lo) | (t > hi), 0.0, a)       # 0 above the max too (types.h:205-213)


def ct_transform(pose14, tbe, t, raw):
    """slerp + lerp interpolation of the begin/end pose at each timestamp, applied to raw."""
    a = alpha(t, tbe[0], tbe[1])
    rots = Rotation.from_quat(np.stack([pose14[0:4], pose14[7:11]]))
    sl = Slerp([0.0, 1.0], rots)
    R = sl(np.clip(a, 0.0, 1.0))
    tr = (1 - a)[:, None] * pose14[4:7] + a[:, None] * pose14[11:14]
    return R.apply(raw) + tr


def radius_search(map_points, resolution, nb, radius, query, k):
    """k nearest map points inside the (2nb+1)^3 voxel cube around the query's voxel and within `radius`,
    FARTHEST FIRST (map.h:449-514)."""
    vq = voxel_coords(query, resolution)
    vm = voxel_coords(map_points, resolution)
    in_cube = np.all(np.abs(vm - vq) <= nb, axis=1)
    cand = map_points[in_cube]
    d = np.linalg.norm(cand - query, axis=1)
    keep = d <= radius
    cand, d = cand[keep], d[keep]
    order = np.argsort(d, kind="stable")[:k]
    return cand[order[::-1]]


def neighborhood(points):
    """mean, un-centred covariance, SVD normal (last right-singular vector) and a2D (neighborhood.h:225-316)."""
    n = len(points)
    mu = points.sum(axis=0) / n
    C = (points[:, :, None] * points[:, None, :]).sum(axis=0) / n - np.outer(mu, mu)
    U, S, Vt = np.linalg.svd(C)
    normal = Vt[2]
    S = np.abs(S)
    a2d = (np.sqrt(S[1]) - np.sqrt(S[2])) / np.sqrt(S[0])
    return normal, a2d


def gn_accumulate(map_points, resolution, nb, radius, raw, world, t, pose14, tbe, k=20, min_nb=20, max_dist=0.3):
    """A (12x12), b (12), n_used and the per-keypoint intermediates, exactly as ct_icp.cpp:753-857 defines them."""
    A = np.zeros((12, 12))
    b = np.zeros(12)
    n_used = 0
    Rb = Rotation.from_quat(pose14[0:4])
    Re = Rotation.from_quat(pose14[7:11])
    tb = pose14[4:7]
    al = alpha(t, tbe[0], tbe[1])
    info = dict(n_neighbors=[], normal=[], a2d=[], farthest=[], used=[])
    for i in range(len(t)):
        p = world[i]
        nbrs = radius_search(map_points, resolution, nb, radius, p, k)
        info["n_neighbors"].append(len(nbrs))
        if len(nbrs) < max(min_nb, 5):
            info["normal"].append(np.zeros(3)); info["a2d"].append(0.0); info["farthest"].append(np.zeros(3))
            info["used"].append(False)
            continue
        normal, a2d = neighborhood(nbrs)
        if normal @ (tb - p) < 0:
            normal = -normal
        q = nbrs[0]                                  # "closest_point" = points[0] = the farthest kept neighbour
        info["normal"].append(normal); info["a2d"].append(a2d); info["farthest"].append(q)
        d = normal @ (p - q)
        if not abs(d) < max_dist:
            info["used"].append(False)
            continue
        w = a2d * a2d
        m = w * normal
        r = m @ (p - q)
        a_ = Rb.apply(raw[i])
        e_ = Re.apply(raw[i])
        u = np.concatenate([(1 - al[i]) * np.cross(a_, m), (1 - al[i]) * m, al[i] * np.cross(e_, m), al[i] * m])
        A += np.outer(u, u)
        b -= u * r
        n_used += 1
        info["used"].append(True)
    for key in info:
        info[key] = np.array(info[key])
    return A, b, n_used, info


def gn_solve_update(A, b, n_used, pose14, prior=None):
    """Normalise, motion prior, solve, Euler increment, pose update (ct_icp.cpp:877-962)."""
    A = A / n_used
    b = b / n_used
    pose14 = np.asarray(pose14, float).copy()
    if prior is not None:
        bc, be, prev_b, prev_e = prior
        A[3:6, 3:6] += bc * np.eye(3)
        b[3:6] -= bc * (pose14[4:7] - pose14[11:14])
        A[9:12, 9:12] += be * np.eye(3)
        b[9:12] -= be * (pose14[11:14] - pose14[4:7] - prev_e + prev_b)
    x = np.linalg.solve(A, b)
    # Rz(gamma) Ry(beta) Rx(alpha): extrinsic xyz Euler angles (alpha, beta, gamma)
    dRb = Rotation.from_euler("xyz", x[0:3])
    dRe = Rotation.from_euler("xyz", x[6:9])
    qb = (dRb * Rotation.from_quat(pose14[0:4])).as_quat()
    qe = (dRe * Rotation.from_quat(pose14[7:11])).as_quat()
    # scipy may return -q; the reference's matrix->quaternion branch yields w >= 0 when the trace is positive
    if qb[3] < 0: qb = -qb
    if qe[3] < 0: qe = -qe
    pose14[0:4], pose14[7:11] = qb, qe
    pose14[4:7] += x[3:6]
    pose14[11:14] += x[9:12]
    return pose14, x


# ---------------------------------------------------------------------------------------------------------------------
# robust-loss route: closed-form derivative of the continuous-time point-to-plane residual
# (reference include/ct_icp/cost_functions.h:46-58,200-225; Ceres differentiates it automatically)
# ---------------------------------------------------------------------------------------------------------------------
def _qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def ct_point_to_plane_analytic(pose14, a, raw, ref, normal, weight):
    """r = w n.(ref - (R(a) raw + (1-a) tb + a te)) and dr/d(tangent), tangent = [begin_quat, end_quat, begin_t, end_t]
    of Ceres' EigenQuaternionParameterization (q <- [sin|d|/|d| d, cos|d|] (x) q, i.e. a LEFT rotation by 2 d).

    With R_b^T R_e = Exp(theta u), R(a) = R_b Exp(a theta u):
        d p / d w_b = -[R raw]x + R [raw]x W_b R_b^T,      W_b = a J_r(a phi) J_l^-1(phi)
        d p / d w_e = -R [raw]x W_e R_e^T,                 W_e = a J_r(a phi) J_r^-1(phi)
    and both W are  a u u^T + s (cos(psi) (I - u u^T) + sin(psi) [u]x),  s = sin(a theta/2) / sin(theta/2),
    psi_e = (1 - a) theta / 2, psi_b = -(1 + a) theta / 2.
    """
    pose14 = np.asarray(pose14, float)
    raw, ref, normal = (np.asarray(v, float) for v in (raw, ref, normal))
    qb = pose14[0:4] / np.linalg.norm(pose14[0:4])
    qe = pose14[7:11] / np.linalg.norm(pose14[7:11])
    Rb, Re = Rotation.from_quat(qb), Rotation.from_quat(qe)
    qrel = _qmul(np.array([-qb[0], -qb[1], -qb[2], qb[3]]), qe)
    if qrel[3] < 0:
        qrel = -qrel
    half = np.arctan2(np.linalg.norm(qrel[:3]), qrel[3])          # theta / 2
    if half < 1e-12:
        u, s, psi_e, psi_b = np.array([1.0, 0, 0]), a, 0.0, 0.0
    else:
        u = qrel[:3] / np.linalg.norm(qrel[:3])
        s = np.sin(a * half) / np.sin(half)
        psi_e, psi_b = (1 - a) * half, -(1 + a) * half
    R = Rb * Rotation.from_rotvec(2 * a * half * u)
    p = R.apply(raw) + (1 - a) * pose14[4:7] + a * pose14[11:14]
    m = weight * normal
    r = m @ (ref - p)
    arot = R.apply(raw)
    c = np.cross(R.inv().apply(m), raw)

    def wt(psi):                                                     # W^T c
        uc = u @ c
        return a * u * uc + s * (np.cos(psi) * (c - u * uc) - np.sin(psi) * np.cross(u, c))

    J = np.zeros(12)
    J[0:3] = -2.0 * (np.cross(arot, m) + Rb.apply(wt(psi_b)))
    J[3:6] = 2.0 * Re.apply(wt(psi_e))
    J[6:9] = -(1 - a) * m
    J[9:12] = -a * m
    return r, J


def quat_plus(q, d):
    """EigenQuaternionParameterization::Plus."""
    n = np.linalg.norm(d)
    if n == 0:
        return np.array(q, float)
    dq = np.concatenate([np.sin(n) / n * np.asarray(d, float), [np.cos(n)]])
    return _qmul(dq, np.asarray(q, float))


def pose_plus(pose14, delta12):
    out = np.array(pose14, float)
    out[0:4] = quat_plus(pose14[0:4], delta12[0:3])
    out[7:11] = quat_plus(pose14[7:11], delta12[3:6])
    out[4:7] += delta12[6:9]
    out[11:14] += delta12[9:12]
    return out
