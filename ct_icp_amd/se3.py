"""Host-side SE(3) helpers with the reference's (Eigen 3) semantics, vectorised in NumPy.

Quaternions are (x, y, z, w) — Eigen's coeffs() order and the order of slam::TSE3::operator[]
(reference include/SlamCore/types.h:378-385). A pose is 7 doubles (qx, qy, qz, qw, tx, ty, tz).
These run on the host only (frame initialisation, synthetic data, tests); the device has its own copies
in ct_icp_amd/csrc/ctgn_device_math.hpp.
"""
from __future__ import annotations

import numpy as np

_ONE = 1.0 - np.finfo(np.float64).eps


def identity_pose14():
    """begin | end pose, both the identity."""
    return np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0] * 2)


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def quat_rotate(q, v):
    """Eigen `q * v` (QuaternionBase::_transformVector): uv = 2 q_v x v; v + w uv + q_v x uv."""
    q = np.asarray(q, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    qv = q[..., :3]
    uv = 2.0 * np.cross(qv, v)
    return v + q[..., 3:4] * uv + np.cross(qv, uv)


def quat_mul(a, b):
    ax, ay, az, aw = np.moveaxis(np.asarray(a, dtype=np.float64), -1, 0)
    bx, by, bz, bw = np.moveaxis(np.asarray(b, dtype=np.float64), -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def quat_conj(q):
    q = np.asarray(q, dtype=np.float64)
    return q * np.array([-1.0, -1.0, -1.0, 1.0])


def quat_slerp(a, b, t):
    """Eigen QuaternionBase::slerp(t, other) for one (a, b) pair and an array of t. Not re-normalised."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    d = float(np.dot(a, b))
    ad = abs(d)
    if ad >= _ONE:
        s0, s1 = 1.0 - t, t
    else:
        th = np.arccos(ad)
        st = np.sin(th)
        s0 = np.sin((1.0 - t) * th) / st
        s1 = np.sin(t * th) / st
    if d < 0:
        s1 = -s1
    return s0[..., None] * a + s1[..., None] * b


def quat_to_matrix(q):
    x, y, z, w = np.asarray(q, dtype=np.float64)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def matrix_to_quat(m):
    """Eigen Quaterniond(Matrix3d): trace branch, else largest-diagonal branch (first-max tie rule)."""
    m = np.asarray(m, dtype=np.float64)
    q = np.zeros(4)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def quat_from_rotvec(rv):
    rv = np.asarray(rv, dtype=np.float64)
    th = np.linalg.norm(rv)
    if th < 1e-300:
        return np.array([0.0, 0.0, 0.0, 1.0])
    ax = rv / th
    return np.concatenate([ax * np.sin(0.5 * th), [np.cos(0.5 * th)]])


def alpha_timestamp(t, t_begin, t_end):
    """TPose::GetAlphaTimestamp (reference include/SlamCore/types.h:192-219), incl. the 0-above-max quirk."""
    t = np.asarray(t, dtype=np.float64)
    lo, hi = min(t_begin, t_end), max(t_begin, t_end)
    if lo == hi:
        a = np.ones_like(t)
    else:
        a = (t - lo) / (hi - lo)
    return np.where((t < lo) | (t > hi), 0.0, a)


def ct_transform(pose14, t_begin_end, t, raw):
    """pose_begin.InterpolatePose(pose_end, t) * raw for arrays (types.h:453-470, :360-366, :353-357)."""
    pose14 = np.asarray(pose14, dtype=np.float64).ravel()
    raw = np.asarray(raw, dtype=np.float64).reshape(-1, 3)
    a = alpha_timestamp(t, t_begin_end[0], t_begin_end[1])
    q = quat_normalize(quat_slerp(pose14[0:4], pose14[7:11], a))
    tr = (1.0 - a)[:, None] * pose14[4:7] + a[:, None] * pose14[11:14]
    return quat_rotate(q, raw) + tr


def angular_distance_deg(qa, qb):
    """slam::AngularDistance (types.h:141-150) on two unit quaternions, in degrees."""
    Ra, Rb = quat_to_matrix(qa), quat_to_matrix(qb)
    c = (np.trace(Ra @ Rb.T) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def pose_error(pose_a, pose_b):
    """(max translation error in m, max rotation error in rad) over the begin and end poses of two 14-vectors."""
    a = np.asarray(pose_a, dtype=np.float64).ravel()
    b = np.asarray(pose_b, dtype=np.float64).ravel()
    tr = max(np.linalg.norm(a[4:7] - b[4:7]), np.linalg.norm(a[11:14] - b[11:14]))
    # angle of the relative rotation from the vector part of conj(qa) * qb: 2 atan2(|v|, |w|) -- accurate down to 1e-16, where the
    # acos((trace - 1) / 2) of AngularDistance (types.h:141-150) stops resolving at sqrt(eps) ~ 1.5e-8
    def _angle(qa, qb):
        qa, qb = quat_normalize(qa), quat_normalize(qb)
        w = float(np.dot(qa, qb))
        v = qa[3] * qb[0:3] - qb[3] * qa[0:3] - np.cross(qa[0:3], qb[0:3])
        return 2.0 * np.arctan2(np.linalg.norm(v), abs(w))
    rot = max(_angle(a[0:4], b[0:4]), _angle(a[7:11], b[7:11]))
    return float(tr), float(rot)
