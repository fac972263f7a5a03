"""GPU parity tests of the robust-loss (CERES-profile) route: ctgn_solve_robust / ctgn_register_robust
through the C ABI against oracle/ctgn_oracle_robust.c on the same seeded inputs.

The two sides share no arithmetic for the derivative (closed form on the GPU, forward-mode jets in the oracle), the
12x12 solve (pivoted LDL^T vs Cholesky) or the reductions (fixed block tree vs serial), so agreement is FP64 round-off
amplified by the conditioning of the normal equations: asserted at 1e-8 .. 1e-6, far inside the contractual
1e-4 m / 1e-4 rad."""
import numpy as np
import pytest

import ct_icp_amd as cia
from ct_icp_amd import _lib as L
from ct_icp_amd import se3, synthetic as syn
from oracle import oracle as orc
from conftest import build_maps

pytestmark = pytest.mark.gpu


def _setup(case, frame, voxel, n_map=None, perturb=(0.004, 0.04), seed=3):
    om, gm = build_maps(case, n_map if n_map is not None else frame, with_gpu=True)
    sc = case["scans"][frame]
    sel = syn.grid_sample_indices(sc.raw, voxel)
    raw, t = sc.raw[sel], sc.t[sel]
    pose0 = syn.perturb_pose(sc.pose_gt, perturb[0], perturb[1], seed=seed)
    return om, gm, sc, raw, t, pose0


def _opts(**kw):
    d = dict(solver=cia.CERES, debug_print=False, min_number_neighbors=10)
    d.update(kw)
    return cia.CTICPOptions(**d)


def _oopts(o: cia.CTICPOptions):
    return orc.RobustOptions(o.num_iters_icp, o.min_number_neighbors, o.max_number_neighbors, False, o.max_num_residuals,
                             o.loss_function, o.ls_max_num_iters, o.num_closest_neighbors, o.weight_alpha, o.weight_neighborhood,
                             o.power_planarity, o.max_dist_to_plane_ct_icp, o.ls_sigma, o.ls_tolerant_min_threshold,
                             o.threshold_orientation_norm, o.threshold_translation_norm)


def _priors(case, frame, **betas):
    k = case["knots"]
    mm = cia.PreviousFrameMotionModel(**betas)
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(np.concatenate([k[frame - 1], k[frame]]), 0.0, 0.0)
    op = orc.RobustPrior(previous_begin_tr=tuple(k[frame - 1, 4:7]), previous_end_tr=tuple(k[frame, 4:7]),
                         previous_end_quat=tuple(k[frame, 0:4]),
                         beta_location_consistency=mm.beta_location_consistency,
                         beta_constant_velocity=mm.beta_constant_velocity, beta_small_velocity=mm.beta_small_velocity,
                         beta_orientation_consistency=mm.beta_orientation_consistency)
    return mm, op


@pytest.mark.parametrize("case_name,voxel,loss", [("box_case", 0.4, "CAUCHY"), ("street_case", 0.6, "HUBER"),
                                                  ("box_case", 0.5, "TOLERANT"), ("street_case", 0.8, "TRUNCATED"),
                                                  ("box_case", 0.5, "STANDARD")])
def test_blocks_and_normal_equations_match_oracle(case_name, voxel, loss, request):
    """One ICP iteration with a zero-iteration inner solve = Ceres' "iteration 0": blocks, weights, cost, J^T J, J^T r."""
    case = request.getfixturevalue(case_name)
    om, gm, sc, raw, t, pose0 = _setup(case, 5, voxel)
    mm, op = _priors(case, 5, beta_small_velocity=0.002, beta_orientation_consistency=0.003)
    o = _opts(num_iters_icp=1, ls_max_num_iters=0, loss_function=loss, ls_sigma=0.08, ls_tolerant_min_threshold=0.02)
    s = cia.GnSolver(gm)
    s.set_keypoints(raw, np.zeros_like(raw), t)
    pose1, summ, _ = s.solve_robust(pose0, sc.t_begin_end, o, mm)
    q0 = pose0.copy()
    q0[0:4] /= np.linalg.norm(q0[0:4]); q0[7:11] /= np.linalg.norm(q0[7:11])
    assert np.allclose(pose1, q0, atol=1e-15)                              # no inner iteration: pose untouched
    world = orc.transform_points(q0, sc.t_begin_end, t, raw)
    assert np.abs(s.world_points() - world).max() < 1e-12
    want = orc.robust_build(om, raw, world, t, sc.t_begin_end, _oopts(o), heap_mode=0)
    got = s.robust_blocks()
    kp = want["keypoint"]
    assert len(kp) > 400 and summ.num_residuals_used == len(kp)
    valid = got["rank"] >= 0
    assert np.array_equal(np.nonzero(valid)[0], kp)
    assert np.array_equal(got["rank"][kp], np.arange(len(kp)))
    assert np.array_equal(got["ref"][kp], want["ref"])                      # bit-exact neighbour sets
    assert np.array_equal(got["alpha"][kp], want["alpha"])
    assert np.abs(got["weight"][kp] - want["weight"]).max() < 1e-9
    sign = np.sign(np.sum(got["normal"][kp] * want["normal"], axis=1))
    planar = want["weight"] > 0.1
    assert np.abs(got["normal"][kp][planar] * sign[planar, None] - want["normal"][planar]).max() < 1e-7
    cost, H, g = orc.robust_evaluate(want, _oopts(o), op, q0)
    rep = s.robust_report()
    assert abs(rep["cost"] - cost) < 1e-9 * cost
    assert np.abs(rep["JtJ"] - H).max() < 1e-8 * np.abs(H).max()
    assert np.abs(rep["Jtr"] - g).max() < 1e-8 * np.abs(g).max()


@pytest.mark.parametrize("ls_iters", [1, 3, 6])
def test_inner_solve_matches_oracle(box_case, ls_iters):
    """One ICP iteration, ls_max_num_iters Levenberg-Marquardt iterations on fixed correspondences."""
    case = box_case
    om, gm, sc, raw, t, pose0 = _setup(case, 5, 0.4)
    mm, op = _priors(case, 5)
    o = _opts(num_iters_icp=1, ls_max_num_iters=ls_iters, ls_sigma=0.05)
    s = cia.GnSolver(gm)
    s.set_keypoints(raw, np.zeros_like(raw), t)
    pose1, summ, _ = s.solve_robust(pose0, sc.t_begin_end, o, mm)
    pose_o, world_o, so = orc.register_robust(om, raw, t, pose0, sc.t_begin_end, _oopts(o), op, heap_mode=0)
    tr, rot = se3.pose_error(pose1, pose_o)
    assert tr < 1e-8 and rot < 1e-8, (tr, rot)
    assert summ.success and so.success and summ.num_residuals_used == so.num_residuals_used
    assert summ.num_iters == so.num_iters
    assert np.abs(s.world_points() - world_o).max() < 1e-7
    rep = s.robust_report()
    assert 1 <= rep["ls_iterations"] <= ls_iters and rep["ls_accepted"] >= 1
    # the solve made progress towards the ground truth
    assert se3.pose_error(pose1, sc.pose_gt)[0] < se3.pose_error(pose0, sc.pose_gt)[0]


@pytest.mark.parametrize("case_name,voxel,loss,prior", [("box_case", 0.4, "CAUCHY", True), ("box_case", 0.5, "HUBER", False),
                                                        ("street_case", 0.6, "CAUCHY", False),
                                                        ("box_case", 0.5, "TRUNCATED", True),
                                                        ("box_case", 0.6, "TOLERANT", False)])
def test_register_robust_matches_oracle(case_name, voxel, loss, prior, request):
    case = request.getfixturevalue(case_name)
    om, gm, sc, raw, t, pose0 = _setup(case, 5, voxel)
    mm, op = _priors(case, 5) if prior else (None, None)
    o = _opts(num_iters_icp=8, ls_max_num_iters=5, loss_function=loss, ls_sigma=0.1, ls_tolerant_min_threshold=0.01,
              threshold_orientation_norm=1e-3, threshold_translation_norm=1e-4)
    kp = np.zeros(len(t), dtype=cia.WPOINT3D_DTYPE)
    kp["raw_point"], kp["t"] = raw, t
    frame = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
    summ = cia.CT_ICP_Registration(o).Register(gm, kp, frame, mm)
    pose_o, world_o, so = orc.register_robust(om, raw, t, pose0, sc.t_begin_end, _oopts(o), op, heap_mode=0)
    tr, rot = se3.pose_error(frame.pose14(), pose_o)
    assert summ.success and so.success
    assert tr < 1e-6 and rot < 1e-6, (tr, rot)
    assert summ.num_iters == so.num_iters and summ.num_residuals_used == so.num_residuals_used
    assert np.abs(kp["world_point"] - world_o).max() < 1e-5
    if case_name == "box_case" and loss != "TOLERANT":
        tr_gt, rot_gt = se3.pose_error(frame.pose14(), sc.pose_gt)
        tr0, _ = se3.pose_error(pose0, sc.pose_gt)
        assert tr_gt < 0.5 * tr0 and tr_gt < 0.02, (tr_gt, tr0)


def test_fused_evaluation_and_step_changes_the_summation_order_only(box_case, street_case):
    """tuning robust_fuse: the inner solver's evaluation + step as ONE launch of one block (k_robust_eval_step; the default up to 1 024
    keypoints) against the two-kernel form (k_robust_eval over many blocks + k_robust_step). Same residual blocks, same weights, same
    accepted steps and iteration counts; the normal equations are summed in another fixed order, so the poses agree to rounding, not bit
    for bit — on a frame below the size switch and on one above it."""
    from ct_icp_amd import _lib as L
    try:
        for case, voxel in ((box_case, 0.5), (street_case, 0.45)):
            om, gm, sc, raw, t, pose0 = _setup(case, 5, voxel)
            o = _opts(num_iters_icp=6, ls_max_num_iters=4, loss_function="CAUCHY", threshold_orientation_norm=1e-5, threshold_translation_norm=1e-6)
            got = []
            for fuse in (0, 1):
                L.lib().ctgn_set_tuning(b"robust_fuse", float(fuse))
                kp = np.zeros(len(t), dtype=cia.WPOINT3D_DTYPE)
                kp["raw_point"], kp["t"] = raw, t
                frame = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
                summ = cia.CT_ICP_Registration(o).Register(gm, kp, frame, None)
                got.append((frame.pose14().copy(), kp["world_point"].copy(), summ))
            (pa, wa, sa), (pb, wb, sb) = got
            assert sa.success and sb.success and sa.num_iters == sb.num_iters and sa.num_residuals_used == sb.num_residuals_used, (len(t), sa, sb)
            tr, rot = se3.pose_error(pa, pb)
            assert tr < 1e-9 and rot < 1e-9, (len(t), tr, rot)
            assert np.abs(wa - wb).max() < 1e-8
    finally:
        L.lib().ctgn_set_tuning(b"robust_fuse", -1.0)


def test_residual_cap_and_several_closest_neighbors(box_case):
    case = box_case
    om, gm, sc, raw, t, pose0 = _setup(case, 5, 0.4)
    s = cia.GnSolver(gm)
    s.set_keypoints(raw, np.zeros_like(raw), t)
    for kw in (dict(max_num_residuals=300), dict(num_closest_neighbors=3), dict(num_closest_neighbors=2, max_num_residuals=501)):
        o = _opts(num_iters_icp=2, ls_max_num_iters=3, **kw)
        pose1, summ, _ = s.solve_robust(pose0, sc.t_begin_end, o)
        pose_o, _, so = orc.register_robust(om, raw, t, pose0, sc.t_begin_end, _oopts(o), None, heap_mode=0)
        assert summ.num_residuals_used == so.num_residuals_used
        if "max_num_residuals" in kw:
            assert summ.num_residuals_used == kw["max_num_residuals"]
        tr, rot = se3.pose_error(pose1, pose_o)
        assert tr < 1e-7 and rot < 1e-7, (kw, tr, rot)


def test_soft_failure_and_errors(box_case):
    case = box_case
    om, gm, sc, raw, t, pose0 = _setup(case, 5, 0.5)
    s = cia.GnSolver(gm)
    far = raw[:40] * 50.0                                                  # nothing of the map near these
    s.set_keypoints(far, np.zeros_like(far), t[:40])
    o = _opts(num_iters_icp=3, ls_max_num_iters=2)
    pose1, summ, _ = s.solve_robust(pose0, sc.t_begin_end, o)
    _, _, so = orc.register_robust(om, far, t[:40], pose0, sc.t_begin_end, _oopts(o), None, heap_mode=0)
    assert not summ.success and not so.success
    assert summ.error_log == so.error_log and "not enough keypoints" in summ.error_log
    assert summ.num_residuals_used == so.num_residuals_used
    # a timestamp outside the frame: error instead of the reference's CHECK abort
    t_bad = t[:40].copy()
    t_bad[7] = sc.t_begin_end[1] + 1.0
    s.set_keypoints(raw[:40], np.zeros((40, 3)), t_bad)
    with pytest.raises(cia.CtgnError) as e:
        s.solve_robust(pose0, sc.t_begin_end, o)
    assert e.value.status == L.ERR_TIMESTAMP_RANGE
    s.set_keypoints(raw[:40], np.zeros((40, 3)), t[:40])
    with pytest.raises(cia.CtgnError) as e:
        s.solve_robust(pose0, sc.t_begin_end, _opts(num_closest_neighbors=15, min_number_neighbors=10))
    assert e.value.status == L.ERR_INVALID_ARGUMENT
    # and the GN route still works on the same handle afterwards
    s.set_keypoints(raw, se3.ct_transform(pose0, sc.t_begin_end, t, raw), t)
    _, sg, _ = s.solve(pose0, sc.t_begin_end, cia.CTICPOptions(solver=cia.GN, debug_print=False, min_number_neighbors=10))
    assert sg.success


def test_full_scan_properties(street_case):
    """Size-independent properties on a whole scan (every return a keypoint, no oracle needed): the solve is invariant under a
    permutation of the keypoints (no cap), the cost never rises with more inner iterations, and the world points it leaves are
    the continuous-time transform of the raw points by the pose it returns."""
    case = street_case
    _, gm = build_maps(case, 6, with_gpu=True)
    sc = case["scans"][6]
    raw, t = sc.raw, sc.t
    assert len(t) > 30000
    pose0 = syn.perturb_pose(sc.pose_gt, 0.003, 0.03, seed=11)
    s = cia.GnSolver(gm)
    costs, poses = [], []
    for ls in (0, 1, 2, 4):
        o = _opts(num_iters_icp=1, ls_max_num_iters=ls, min_number_neighbors=20)
        s.set_keypoints(raw, np.zeros_like(raw), t)
        pose, summ, _ = s.solve_robust(pose0, sc.t_begin_end, o)
        rep = s.robust_report()
        costs.append(rep["cost"])
        poses.append(pose)
        assert summ.success and summ.num_residuals_used > 20000
    assert all(c1 <= c0 * (1 + 1e-12) for c0, c1 in zip(costs, costs[1:])), costs
    assert costs[-1] < 0.9 * costs[0]
    world = s.world_points()
    assert np.abs(world - se3.ct_transform(poses[-1], sc.t_begin_end, t, raw)).max() < 1e-9
    perm = np.random.default_rng(3).permutation(len(t))
    s.set_keypoints(raw[perm], np.zeros_like(raw), t[perm])
    pose_p, summ_p, _ = s.solve_robust(pose0, sc.t_begin_end, _opts(num_iters_icp=1, ls_max_num_iters=4, min_number_neighbors=20))
    tr, rot = se3.pose_error(pose_p, poses[-1])
    assert tr < 1e-9 and rot < 1e-9, (tr, rot)
    assert summ_p.num_residuals_used == summ.num_residuals_used


def test_golden_vectors_through_the_gpu(golden, golden_robust):
    """tests/golden/robust_small.npz (NumPy/SciPy derivation, no oracle involved): blocks, weights, cost, normal equations."""
    g, gr = golden, golden_robust
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(float(g["resolution"]), float(g["min_dist"]),
                                                                                  int(g["max_pts"]))], default_radius=float(g["radius"])))
    gm.InsertPointCloud(g["insert_points"])
    betas = gr["betas"]
    mm = cia.PreviousFrameMotionModel(beta_location_consistency=betas[0], beta_constant_velocity=betas[1], beta_small_velocity=betas[2],
                                      beta_orientation_consistency=betas[3])
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(np.concatenate([[0, 0, 0, 1], gr["prev_b"], gr["prev_q"], gr["prev_e"]]), 0.0, 0.0)
    o = cia.CTICPOptions(solver=cia.CERES, debug_print=False, num_iters_icp=1, ls_max_num_iters=0, min_number_neighbors=int(g["min_nb"]),
                         max_number_neighbors=int(g["k"]), loss_function="CAUCHY", ls_sigma=float(gr["sigma"]),
                         weight_alpha=float(gr["weight_alpha"]), weight_neighborhood=float(gr["weight_neighborhood"]),
                         power_planarity=float(gr["power_planarity"]), max_dist_to_plane_ct_icp=float(g["max_dist"]))
    s = cia.GnSolver(gm)
    s.set_keypoints(g["raw"], np.zeros_like(g["raw"]), g["t"])
    _, summ, _ = s.solve_robust(gr["pose0"], g["tbe"], o, mm)
    got = s.robust_blocks()
    kp = gr["keypoint"]
    assert summ.num_residuals_used == len(kp) and np.array_equal(np.nonzero(got["rank"] >= 0)[0], kp)
    assert np.array_equal(got["ref"][kp], gr["ref"])
    assert np.abs(got["weight"][kp] - gr["weight"]).max() < 1e-10
    rep = s.robust_report()
    assert abs(rep["cost"] - gr["cost"]) < 1e-9 * gr["cost"]
    assert np.abs(rep["JtJ"] - gr["JtJ"]).max() < 1e-8 * np.abs(gr["JtJ"]).max()
    assert np.abs(rep["Jtr"] - gr["Jtr"]).max() < 1e-8 * np.abs(gr["Jtr"]).max()
    # and a full inner solve lands on SciPy's minimum of the same fixed blocks
    o.ls_max_num_iters = 50
    s.set_keypoints(g["raw"], np.zeros_like(g["raw"]), g["t"])
    pose, _, _ = s.solve_robust(gr["pose0"], g["tbe"], o, mm)
    assert s.robust_report()["cost"] <= float(gr["cost_opt"]) * (1 + 1e-5)
    tr, rot = se3.pose_error(pose, gr["pose_opt"])
    assert tr < 2e-3 and rot < 1e-3, (tr, rot)          # flat valley along the regularised directions; Ceres stops at 1e-6 relative
