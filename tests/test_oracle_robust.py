"""Pins for the robust-loss (CERES-profile) oracle (oracle/ctgn_oracle_robust.c).

The reference's own tests for this route are test/unit/ct_icp/test_cost_functions.cxx:70-105 (the CT functor leaves a
point that already lies on the plane at zero cost and a Ceres solve pulls a perturbed pose back) and nothing else; Ceres
itself is absent. So the restatement is pinned by: finite differences and an independent closed form for the Jacobian,
numerical derivatives for the loss functions, SciPy's own robust least squares for the minimiser, and ground-truth
recovery on synthetic scans."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from oracle import numpy_check as npc
from oracle import oracle as orc
from ct_icp_amd import se3, synthetic as syn
from conftest import build_maps


def _rand_pose(rng, rot=0.3, tr=1.0, rel_rot=0.1):
    qb = se3.quat_from_rotvec(rng.normal(size=3) * rot)
    qe = se3.quat_mul(se3.quat_from_rotvec(rng.normal(size=3) * rel_rot), qb)
    return np.concatenate([qb, rng.normal(size=3) * tr, qe, rng.normal(size=3) * tr])


def test_jet_jacobian_matches_finite_differences_and_closed_form():
    rng = np.random.default_rng(0)
    for trial in range(40):
        pose = _rand_pose(rng, rel_rot=[0.0, 1e-9, 1e-3, 0.1, 1.0][trial % 5])
        a = [0.0, 1.0, 0.5, rng.uniform(), rng.uniform()][trial % 5]
        raw, ref = rng.normal(size=3) * 10, rng.normal(size=3) * 10
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        w = rng.uniform(0.1, 1.0)
        r, J = orc.ct_point_to_plane(pose, a, raw, ref, n, w)
        r2, J2 = npc.ct_point_to_plane_analytic(pose, a, raw, ref, n, w)
        assert abs(r - r2) < 1e-11 * (1 + abs(r))
        assert np.allclose(J, J2, rtol=1e-9, atol=1e-9), (trial, J, J2)
        h = 1e-6
        Jfd = np.zeros(12)
        for k in range(12):
            d = np.zeros(12)
            d[k] = h
            rp = orc.ct_point_to_plane(npc.pose_plus(pose, d), a, raw, ref, n, w, jacobian=False)
            rm = orc.ct_point_to_plane(npc.pose_plus(pose, -d), a, raw, ref, n, w, jacobian=False)
            Jfd[k] = (rp - rm) / (2 * h)
        assert np.allclose(J, Jfd, rtol=1e-6, atol=1e-6), (trial, J, Jfd)


def test_point_on_plane_has_zero_residual():
    # test_cost_functions.cxx:70-84: the functor evaluated at the true pose on an on-plane point returns 0
    rng = np.random.default_rng(1)
    pose = _rand_pose(rng)
    tbe = (0.0, 1.0)
    raw = rng.normal(size=(20, 3)) * 5
    t = rng.uniform(size=20)
    world = orc.transform_points(pose, tbe, t, raw)
    for i in range(20):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        tangent = np.cross(n, rng.normal(size=3))
        r = orc.ct_point_to_plane(pose, t[i], raw[i], world[i] + tangent, n, 0.8, jacobian=False)
        assert abs(r) < 1e-12


@pytest.mark.parametrize("kind", ["STANDARD", "CAUCHY", "HUBER", "TOLERANT", "TRUNCATED"])
def test_loss_derivatives(kind):
    sigma, tol = 0.1, 0.05
    for s in [0.0, 1e-4, 0.005, 0.0099, 0.0101, 0.05, 0.5, 3.0, 50.0]:
        rho = orc.loss_evaluate(kind, sigma, tol, s)
        h = 1e-7 * max(1.0, s)
        if kind in ("HUBER", "TRUNCATED") and abs(s - sigma * sigma) < 2 * h:
            continue
        lo = s - h if s > h else s
        d1 = (orc.loss_evaluate(kind, sigma, tol, s + h)[0] - orc.loss_evaluate(kind, sigma, tol, lo)[0]) / (s + h - lo)
        assert abs(d1 - rho[1]) < 1e-5 * (1 + abs(rho[1])), (s, d1, rho)
        d2 = (orc.loss_evaluate(kind, sigma, tol, s + h)[1] - orc.loss_evaluate(kind, sigma, tol, lo)[1]) / (s + h - lo)
        assert abs(d2 - rho[2]) < 1e-4 * (1 + abs(rho[2])), (s, d2, rho)
    # published closed forms
    assert np.allclose(orc.loss_evaluate("CAUCHY", 0.1, 0, 0.02), [0.01 * np.log(3.0), 1 / 3, -100 / 9])
    assert np.allclose(orc.loss_evaluate("HUBER", 0.1, 0, 0.04), [2 * 0.1 * 0.2 - 0.01, 0.5, -0.5 / 0.08])
    assert np.allclose(orc.loss_evaluate("TRUNCATED", 0.1, 0, 0.04), [0.01, 0, 0])
    assert orc.loss_evaluate("TOLERANT", 0.1, 0.05, 0.0)[0] == pytest.approx(0.0, abs=1e-15)


def _synthetic_blocks(rng, n, pose_gt, outliers=0.0, noise=0.0):
    """Planes seen from a moving sensor: points on a few planes, residual blocks against the true planes."""
    planes = [(np.array([1.0, 0, 0]), 10.0), (np.array([0, 1.0, 0]), -8.0), (np.array([0, 0, 1.0]), -1.5),
              (np.array([0.6, 0.8, 0]), 12.0), (np.array([0, 0.6, 0.8]), 9.0), (np.array([-0.8, 0, 0.6]), 7.0)]
    raw, ref, nrm, alpha = [], [], [], []
    for i in range(n):
        nv, d = planes[i % len(planes)]
        a = rng.uniform()
        # world point on the plane, then back into the sensor frame of time a
        w = rng.normal(size=3) * 6
        w = w - (nv @ w - d) * nv
        q = se3.quat_slerp(pose_gt[0:4], pose_gt[7:11], a)
        tr = (1 - a) * pose_gt[4:7] + a * pose_gt[11:14]
        rw = se3.quat_rotate(se3.quat_conj(q), w - tr)
        anchor = w + np.cross(nv, rng.normal(size=3))           # another point of the same plane
        if rng.uniform() < outliers:
            anchor = anchor + nv * rng.uniform(0.5, 2.0) * rng.choice([-1, 1])
        raw.append(rw + rng.normal(size=3) * noise)
        ref.append(anchor)
        nrm.append(nv)
        alpha.append(a)
    return dict(raw=np.array(raw), ref=np.array(ref), normal=np.array(nrm), weight=rng.uniform(0.3, 1.0, n),
                alpha=np.array(alpha))


def _pose_err(p, q):
    tr, rot = se3.pose_error(p, q)
    return np.degrees(rot), tr


def test_lm_recovers_pose_and_cost_is_monotone():
    # test_cost_functions.cxx:86-105: a solve started from a perturbed pose returns to the true one
    rng = np.random.default_rng(2)
    gt = _rand_pose(rng, rot=0.2, tr=2.0, rel_rot=0.05)
    blocks = _synthetic_blocks(rng, 400, gt)
    start = npc.pose_plus(gt, rng.normal(size=12) * np.array([0.01] * 6 + [0.1] * 6))
    opts = orc.RobustOptions(loss_function="STANDARD")
    costs = [orc.robust_evaluate(blocks, opts, None, start, jacobian=False)]
    pose = start
    for it in range(1, 8):
        pose_k, rep = orc.robust_solve_fixed(blocks, opts, None, start, it)
        costs.append(rep["final_cost"])
    assert all(c1 <= c0 * (1 + 1e-12) for c0, c1 in zip(costs, costs[1:]))
    pose, rep = orc.robust_solve_fixed(blocks, opts, None, start, 50)
    rot, tr = _pose_err(pose, gt)
    assert rep["termination"] == 1 and rot < 1e-6 and tr < 1e-6, (rep, rot, tr)


def test_tolerant_loss_solve_decreases_cost():
    # TolerantLoss is flat below its threshold and linear above: not an outlier rejector, but the corrector's
    # rho'' > 0 branch is only exercised by it
    rng = np.random.default_rng(7)
    gt = _rand_pose(rng, rot=0.2, tr=2.0, rel_rot=0.05)
    blocks = _synthetic_blocks(rng, 300, gt, noise=0.01)
    start = npc.pose_plus(gt, rng.normal(size=12) * np.array([0.01] * 6 + [0.2] * 6))
    opts = orc.RobustOptions(loss_function="TOLERANT", ls_sigma=0.02, ls_tolerant_min_threshold=0.01)
    pose, rep = orc.robust_solve_fixed(blocks, opts, None, start, 50)
    assert rep["termination"] >= 0 and rep["final_cost"] < 0.05 * rep["initial_cost"]
    assert _pose_err(pose, gt)[1] < 0.5 * _pose_err(start, gt)[1]


@pytest.mark.parametrize("kind", ["CAUCHY", "HUBER", "TRUNCATED"])
def test_robust_losses_reject_outliers(kind):
    rng = np.random.default_rng(3)
    gt = _rand_pose(rng, rot=0.2, tr=2.0, rel_rot=0.05)
    blocks = _synthetic_blocks(rng, 600, gt, outliers=0.15, noise=0.002)
    start = npc.pose_plus(gt, rng.normal(size=12) * np.array([0.002] * 6 + [0.02] * 6))
    plain, _ = orc.robust_solve_fixed(blocks, orc.RobustOptions(loss_function="STANDARD"), None, start, 50)
    robust, rep = orc.robust_solve_fixed(blocks, orc.RobustOptions(loss_function=kind, ls_sigma=0.05), None, start, 50)
    assert rep["termination"] >= 0
    assert _pose_err(robust, gt)[1] < 0.3 * _pose_err(plain, gt)[1], (_pose_err(robust, gt), _pose_err(plain, gt))


@pytest.mark.parametrize("with_prior", [False, True])
def test_minimiser_agrees_with_scipy(with_prior):
    """Same cost, different solver and different arithmetic: SciPy's trust-region-reflective least squares on
    1/2 sum sigma^2 log(1 + r^2 / sigma^2) (Ceres' CauchyLoss(sigma)) + the regularisers must land on the same pose."""
    rng = np.random.default_rng(4)
    gt = _rand_pose(rng, rot=0.2, tr=2.0, rel_rot=0.05)
    blocks = _synthetic_blocks(rng, 500, gt, outliers=0.1, noise=0.003)
    sigma = 0.05
    opts = orc.RobustOptions(loss_function="CAUCHY", ls_sigma=sigma)
    prior = None
    if with_prior:
        prior = orc.RobustPrior(beta_location_consistency=0.01, beta_constant_velocity=0.02, beta_small_velocity=0.005,
                                beta_orientation_consistency=0.03, previous_begin_tr=tuple(gt[4:7] - 0.3),
                                previous_end_tr=tuple(gt[4:7] + 0.01), previous_end_quat=tuple(gt[0:4]))
    start = npc.pose_plus(gt, rng.normal(size=12) * np.array([0.002] * 6 + [0.02] * 6))
    pose, rep = orc.robust_solve_fixed(blocks, opts, prior, start, 200)
    n = len(blocks["weight"])

    def point_res(d):                       # vectorised, SciPy Rotation / Slerp: shares nothing with the C oracle
        p = npc.pose_plus(start, d)
        world = npc.ct_transform(p, (0.0, 1.0), blocks["alpha"], blocks["raw"])
        return blocks["weight"] * np.sum(blocks["normal"] * (blocks["ref"] - world), axis=1)

    def reg_res(d):
        if prior is None:
            return np.zeros(0)
        p = npc.pose_plus(start, d)
        out = []
        out += list(np.sqrt(n * prior.beta_location_consistency) * (p[4:7] - np.array(prior.previous_end_tr)))
        sc = p[0:4] @ np.array(prior.previous_end_quat)
        out += [np.sqrt(n * prior.beta_orientation_consistency) * (1 - sc * sc)]
        vel = np.array(prior.previous_end_tr) - np.array(prior.previous_begin_tr)
        out += list(np.sqrt(n * prior.beta_constant_velocity) * (p[11:14] - p[4:7] - vel))
        out += list(np.sqrt(n * prior.beta_small_velocity) * (p[4:7] - p[11:14]))
        return np.array(out)

    # 1/2 sum rho(r^2) as a plain least-squares problem: r' = sign(r) sqrt(rho(r^2)); the regularisers carry no loss
    def fun(d):
        r = point_res(d)
        return np.concatenate([np.sign(r) * sigma * np.sqrt(np.log1p(r * r / sigma ** 2)), reg_res(d)])

    sol = least_squares(fun, np.zeros(12), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    ref_pose = npc.pose_plus(start, sol.x)
    cost_ref = orc.robust_evaluate(blocks, opts, prior, ref_pose, jacobian=False)
    assert rep["final_cost"] <= cost_ref * (1 + 1e-5)      # Ceres stops once a step gains < function_tolerance = 1e-6
    # the oracle's gradient (jets + corrector + regularisers) vanishes at SciPy's optimum
    _, H, g = orc.robust_evaluate(blocks, opts, prior, ref_pose)
    assert np.max(np.abs(g) / np.sqrt(np.diag(H))) < 1e-6 * np.sqrt(2 * cost_ref)
    rot, tr = _pose_err(pose, ref_pose)
    if with_prior:      # flat valley along the regularised directions: the 1e-6 function tolerance stops earlier
        assert rot < 0.05 and tr < 2e-3, (rot, tr, rep)
    else:
        assert rot < 1e-4 and tr < 1e-5, (rot, tr, rep)


def test_register_robust_recovers_ground_truth(box_case):
    case = box_case
    om, _ = build_maps(case, 4)
    sc = case["scans"][4]
    idx = syn.grid_sample_indices(sc.raw, 0.6)
    raw, t = sc.raw[idx], sc.t[idx]
    gt = syn.frame_pose14(case["knots"], 4)
    init = syn.perturb_pose(gt, 0.01, 0.06, seed=5)
    opts = orc.RobustOptions(num_iters_icp=15, ls_max_num_iters=5, ls_sigma=0.1, min_number_neighbors=10,
                             threshold_orientation_norm=1e-4, threshold_translation_norm=1e-4)
    pose, world, s = orc.register_robust(om, raw, t, init, tuple(sc.t_begin_end), opts)
    assert s.success and s.num_residuals_used > 100
    rot, tr = _pose_err(pose, gt)
    rot0, tr0 = _pose_err(init, gt)
    assert tr < 0.02 and rot < 0.1 and tr < 0.5 * tr0, (rot, tr, rot0, tr0)
    assert np.allclose(world, orc.transform_points(pose, tuple(sc.t_begin_end), t, raw), atol=1e-12)
    # the residual cap keeps the first max_num_residuals blocks in keypoint order
    capped = orc.robust_build(om, raw, world, t, tuple(sc.t_begin_end), orc.RobustOptions(max_num_residuals=50, min_number_neighbors=10))
    full = orc.robust_build(om, raw, world, t, tuple(sc.t_begin_end), orc.RobustOptions(min_number_neighbors=10))
    assert len(capped["weight"]) == 50 and np.array_equal(capped["keypoint"], full["keypoint"][:50])
    # too few residuals: soft failure with the reference's message
    tiny = orc.RobustOptions(min_number_neighbors=10)
    _, _, s2 = orc.register_robust(om, raw[:3] * 100.0, t[:3], init, tuple(sc.t_begin_end), tiny)
    assert not s2.success and "not enough keypoints" in s2.error_log


# ---------------------------------------------------------------- golden vectors (tests/golden/make_golden_robust.py)
def _golden_setup(golden, golden_robust):
    g, gr = golden, golden_robust
    om = orc.Map(resolutions=[(float(g["resolution"]), float(g["min_dist"]), int(g["max_pts"]))], default_radius=float(g["radius"]))
    om.insert(g["insert_points"])
    betas = gr["betas"]
    opts = orc.RobustOptions(num_iters_icp=1, ls_max_num_iters=0, min_number_neighbors=int(g["min_nb"]), max_number_neighbors=int(g["k"]),
                             loss_function="CAUCHY", ls_sigma=float(gr["sigma"]), weight_alpha=float(gr["weight_alpha"]),
                             weight_neighborhood=float(gr["weight_neighborhood"]), power_planarity=float(gr["power_planarity"]),
                             max_dist_to_plane_ct_icp=float(g["max_dist"]))
    prior = orc.RobustPrior(beta_location_consistency=betas[0], beta_constant_velocity=betas[1], beta_small_velocity=betas[2],
                            beta_orientation_consistency=betas[3], previous_begin_tr=tuple(gr["prev_b"]),
                            previous_end_tr=tuple(gr["prev_e"]), previous_end_quat=tuple(gr["prev_q"]))
    return om, opts, prior


def test_oracle_reproduces_the_numpy_golden_vectors(golden, golden_robust):
    """Blocks, weights, cost and loss-corrected normal equations of the independent NumPy derivation; its SciPy minimiser."""
    g, gr = golden, golden_robust
    om, opts, prior = _golden_setup(g, gr)
    pose0 = gr["pose0"]
    world0 = orc.transform_points(pose0, g["tbe"], g["t"], g["raw"])
    blocks = orc.robust_build(om, g["raw"], world0, g["t"], g["tbe"], opts, heap_mode=0)
    assert np.array_equal(blocks["keypoint"], gr["keypoint"])
    assert np.abs(blocks["ref"] - gr["ref"]).max() == 0.0
    assert np.abs(blocks["weight"] - gr["weight"]).max() < 1e-10
    assert np.abs(blocks["alpha"] - gr["alpha"]).max() < 1e-15
    sign = np.sign(np.sum(blocks["normal"] * gr["normal"], axis=1))
    planar = gr["weight"] > 0.1
    assert np.abs(blocks["normal"][planar] * sign[planar, None] - gr["normal"][planar]).max() < 1e-8
    cost, H, grad = orc.robust_evaluate(blocks, opts, prior, pose0)
    assert abs(cost - gr["cost"]) < 1e-9 * gr["cost"]
    assert np.abs(H - gr["JtJ"]).max() < 1e-8 * np.abs(gr["JtJ"]).max()
    assert np.abs(grad - gr["Jtr"]).max() < 1e-8 * np.abs(gr["Jtr"]).max()
    # the oracle's Levenberg-Marquardt on the same blocks reaches SciPy's minimum; its gradient vanishes there
    pose, rep = orc.robust_solve_fixed(blocks, opts, prior, pose0, 100)
    assert rep["final_cost"] <= float(gr["cost_opt"]) * (1 + 1e-5)
    _, Ho, go = orc.robust_evaluate(blocks, opts, prior, gr["pose_opt"])
    assert np.max(np.abs(go) / np.sqrt(np.diag(Ho))) < 1e-6 * np.sqrt(2 * float(gr["cost_opt"]))
