"""Generates tests/golden/robust_small.npz — golden vectors for the robust-loss (CERES-profile) route, produced by the
INDEPENDENT NumPy/SciPy derivation (oracle/numpy_check.py: brute-force neighbourhoods + SVD normals already stored in
gn_small.npz, the closed-form SO(3) Jacobian, SciPy's trust-region least squares), not by the C oracle and not by the product.

Same scene / map / keypoints / start pose as gn_small.npz. Contents: the residual blocks of the first ICP iteration (weights of
ct_icp.cpp:525-532,574-579), cost = 1/2 sum rho(r^2) + regularisers, the loss-corrected J^T J and J^T r at the start pose
(Cauchy sigma, all four PreviousFrameMotionModel regularisers), and the minimiser SciPy finds on those fixed blocks.

    python tests/golden/make_golden_robust.py        # rewrites robust_small.npz (deterministic)
"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import numpy_check as npc             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = dict(np.load(os.path.join(HERE, "gn_small.npz")))
    min_nb, max_dist = int(g["min_nb"]), float(g["max_dist"])
    sigma, w_alpha, w_nbr, power = 0.07, 0.9, 0.1, 2.0
    betas = dict(loc=0.004, vel=0.003, small=0.002, orient=0.005)
    pose0 = g["pose0"].copy()
    pose0[0:4] /= np.linalg.norm(pose0[0:4])
    pose0[7:11] /= np.linalg.norm(pose0[7:11])
    raw, t, tbe = g["raw"], g["t"], g["tbe"]
    world0 = npc.ct_transform(pose0, tbe, t, raw)
    valid = g["n_neighbors"] >= max(min_nb, 5)                       # ct_icp.cpp:566-567
    kp = np.nonzero(valid)[0]
    lw, ln = w_alpha / (w_alpha + w_nbr), w_nbr / (w_alpha + w_nbr)
    far = g["farthest"][kp]
    weight = lw * g["a2d"][kp] ** power + ln * np.exp(-np.linalg.norm(far - world0[kp], axis=1) / (max_dist * min_nb))
    alpha = npc.alpha(t[kp], tbe[0], tbe[1])
    blocks = dict(raw=raw[kp], ref=far, normal=g["normal"][kp], weight=weight, alpha=alpha)
    n = len(kp)
    prev_b, prev_e = g["prior_prev_b"], g["prior_prev_e"]
    prev_q = np.array([0.01, -0.02, 0.015, 1.0])
    prev_q /= np.linalg.norm(prev_q)

    def point_terms(pose):
        r = np.zeros(n)
        J = np.zeros((n, 12))
        for i in range(n):
            r[i], J[i] = npc.ct_point_to_plane_analytic(pose, alpha[i], blocks["raw"][i], blocks["ref"][i], blocks["normal"][i], weight[i])
        return r, J

    def reg_terms(pose):
        """Residuals and tangent Jacobian of AddConstraintsToCeresProblem (motion_model.cpp:12-61)."""
        rows, res = [], []
        bl, bo, bv, bs = (np.sqrt(n * betas[k]) for k in ("loc", "orient", "vel", "small"))
        for c in range(3):
            J = np.zeros(12); J[6 + c] = bl
            rows.append(J); res.append(bl * (pose[4 + c] - prev_e[c]))
        sc = pose[0:4] @ prev_q
        q = pose[0:4]
        amb = -2.0 * bo * sc * prev_q                                  # d/dq of bo (1 - (q.qp)^2)
        plus = np.array([[q[3], q[2], -q[1]], [-q[2], q[3], q[0]], [q[1], -q[0], q[3]], [-q[0], -q[1], -q[2]]])
        J = np.zeros(12); J[0:3] = amb @ plus
        rows.append(J); res.append(bo * (1.0 - sc * sc))
        for c in range(3):
            J = np.zeros(12); J[6 + c] = -bv; J[9 + c] = bv
            rows.append(J); res.append(bv * (pose[11 + c] - pose[4 + c] - (prev_e[c] - prev_b[c])))
        for c in range(3):
            J = np.zeros(12); J[6 + c] = bs; J[9 + c] = -bs
            rows.append(J); res.append(bs * (pose[4 + c] - pose[11 + c]))
        return np.array(res), np.array(rows)

    # cost and corrected normal equations at pose0 (Cauchy: rho = s2 log(1 + s/s2), rho' = 1/(1 + s/s2), rho'' < 0 => plain IRLS)
    r, J = point_terms(pose0)
    s2 = sigma * sigma
    rho1 = 1.0 / (1.0 + r * r / s2)
    cost = 0.5 * np.sum(s2 * np.log1p(r * r / s2))
    Jc, rc = J * np.sqrt(rho1)[:, None], r * np.sqrt(rho1)
    rr, Jr = reg_terms(pose0)
    cost += 0.5 * rr @ rr
    H = Jc.T @ Jc + Jr.T @ Jr
    grad = Jc.T @ rc + Jr.T @ rr

    def fun(d):
        p = npc.pose_plus(pose0, d)
        world = npc.ct_transform(p, (0.0, 1.0), alpha, blocks["raw"])       # alpha is already the interpolation parameter
        rp = weight * np.sum(blocks["normal"] * (blocks["ref"] - world), axis=1)
        return np.concatenate([np.sign(rp) * sigma * np.sqrt(np.log1p(rp * rp / s2)), reg_terms(p)[0]])

    sol = least_squares(fun, np.zeros(12), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    pose_opt = npc.pose_plus(pose0, sol.x)
    out = os.path.join(HERE, "robust_small.npz")
    np.savez_compressed(out, keypoint=kp, ref=far, normal=blocks["normal"], weight=weight, alpha=alpha, pose0=pose0, sigma=sigma,
                        weight_alpha=w_alpha, weight_neighborhood=w_nbr, power_planarity=power,
                        betas=np.array([betas["loc"], betas["vel"], betas["small"], betas["orient"]]), prev_b=prev_b, prev_e=prev_e,
                        prev_q=prev_q, cost=cost, JtJ=H, Jtr=grad, pose_opt=pose_opt, cost_opt=float(sol.cost))
    print("wrote", out, "blocks", n, "cost", cost, "cost at SciPy's optimum", sol.cost)


if __name__ == "__main__":
    main()
