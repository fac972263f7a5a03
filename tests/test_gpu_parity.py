"""GPU parity tests (run on an MI355X with `pytest -m gpu`): the HIP path, called through the C ABI of libctgn.so,
against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): neighbour sets / indices / gate decisions are bit-exact; floating-point results agree
to the stated SE(3) tolerance of 1e-4 m / 1e-4 rad — the tests assert the much tighter FP64 agreement actually
expected (1e-9 .. 1e-7), so a drift far below the contractual tolerance is still caught.
"""
import numpy as np
import ctypes as C
import pytest

import ct_icp_amd as cia
from ct_icp_amd import _lib as L
from ct_icp_amd import se3, synthetic as syn
from oracle import oracle as orc
from conftest import build_maps

pytestmark = pytest.mark.gpu

POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-4          # the stated tolerance
TIGHT = 1e-8                                   # what FP64 on both sides actually delivers


def _keypoints(case, frame, voxel, n_max=None, perturb=(0.005, 0.03), seed=1):
    sc = case["scans"][frame]
    sel = syn.grid_sample_indices(sc.raw, voxel)
    if n_max:
        sel = sel[:n_max]
    raw, t = sc.raw[sel], sc.t[sel]
    pose0 = syn.perturb_pose(sc.pose_gt, perturb[0], perturb[1], seed=seed)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, t, raw)
    return sc, raw, t, pose0, world0


def _prior(case, frame):
    k = case["knots"]
    mm = cia.PreviousFrameMotionModel()
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(np.concatenate([k[frame - 1], k[frame]]), 0.0, 0.0)
    op = orc.MotionPrior(previous_begin_tr=k[frame - 1, 4:7], previous_end_tr=k[frame, 4:7])
    return mm, op


def _opts(**kw):
    d = dict(solver=cia.GN, debug_print=False)
    d.update(kw)
    return cia.CTICPOptions(**d)


def _oopts(o: cia.CTICPOptions):
    return orc.Options(o.num_iters_icp, o.min_number_neighbors, o.max_number_neighbors, False, o.max_dist_to_plane_ct_icp,
                       o.threshold_orientation_norm)


# ------------------------------------------------------------------------------------------------- neighbour search
@pytest.mark.parametrize("case_name", ["box_case", "street_case"])
def test_radius_search_is_bit_exact(case_name, request):
    case = request.getfixturevalue(case_name)
    om, gm = build_maps(case, 5, with_gpu=True)
    rng = np.random.default_rng(0)
    pts = case["scans"][5].world_gt
    qs = pts[rng.choice(len(pts), 600, replace=False)] + rng.normal(0, 0.05, (600, 3))
    got = gm.ComputeNeighborhoods(qs, 20)
    n_full = 0
    for q, g in zip(qs, got):
        want = om.radius_search(q, 0.0, 20, heap_mode=0)
        assert g.shape == want.shape and np.array_equal(g, want)
        assert np.array_equal(want, om.radius_search(q, 0.0, 20, heap_mode=1))      # no ties in this data: the reference's queue (0) and the total order (1) agree
        n_full += len(g) == 20
    assert n_full > 300
    # 1-NN identity (reference test/unit/SlamCore/test_map.cxx:25-33) through the GPU map
    mp = gm.MapAsPointCloud(0)
    sub = mp[rng.choice(len(mp), 300, replace=False)]
    for q, g in zip(sub, gm.ComputeNeighborhoods(sub, 1)):
        assert len(g) == 1 and np.array_equal(g[0], q)
    # explicit radius selects another sweep width
    r = case["default_radius"] * 0.5
    for q, g in zip(qs[:50], gm.ComputeNeighborhoods(qs[:50], 12, radius=r)):
        assert np.array_equal(g, om.radius_search(q, r, 12, heap_mode=0))


# ------------------------------------------------------------------------------------------------- one accumulation
@pytest.mark.parametrize("variant", [0, 1, 2, 5])
@pytest.mark.parametrize("case_name,voxel", [("box_case", 0.4), ("street_case", 0.6)])
def test_accumulate_matches_oracle(case_name, voxel, variant, request):
    case = request.getfixturevalue(case_name)
    om, gm = build_maps(case, 6, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(case, 6, voxel)
    assert len(t) > 1500
    o = _opts(num_iters_icp=1, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    s.set_variant(variant)
    s.set_debug(True)
    s.set_keypoints(raw, world0, t)
    pose1, summ, _ = s.solve(pose0, sc.t_begin_end, o)
    dbg = s.get_debug()
    A, b, n_used = s.get_system()
    Ao, bo, no, info = orc.gn_accumulate(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), heap_mode=0, debug=True)
    assert np.array_equal(dbg["n_neighbors"], info["n_neighbors"])
    has = info["n_neighbors"] >= max(o.min_number_neighbors, 5)
    assert has.sum() > 500
    assert np.array_equal(dbg["farthest"][has], info["farthest"][has])
    assert np.array_equal(dbg["used"], info["used"])
    assert n_used == no == summ.num_residuals_used
    # round 4: sums, mean and covariance in the reference build's arithmetic (no fused multiply-add). Default solver mode (hybrid): the
    # gate decisions above are the oracle's, normals and a2D to rounding; exact mode (Eigen's JacobiSVD restated, normal_a2d_exact,
    # neighborhood.h:236-244,293-311): a2D and normals are the oracle's bit for bit wherever a neighbourhood is valid
    assert np.abs(dbg["a2d"][has] - info["a2d"][has]).max() < 1e-9
    planar = has & (info["a2d"] > 1e-3)
    assert np.abs(dbg["normal"][planar] - info["normal"][planar]).max() < 1e-9
    s.set_normals(1)
    s.set_keypoints(raw, world0, t)
    s.solve(pose0, sc.t_begin_end, o)
    dbx = s.get_debug()
    s.set_normals(0)
    assert np.array_equal(dbx["used"], info["used"])
    assert np.array_equal(dbx["a2d"][has], info["a2d"][has])
    assert np.array_equal(dbx["normal"][has], info["normal"][has])
    scale = np.abs(Ao).max()
    assert np.abs(A - Ao).max() < 1e-10 * scale and np.abs(b - bo).max() < 1e-10 * max(np.abs(bo).max(), 1e-30) + 1e-14
    # and the solve that followed
    pose_o, x_o, _ = orc.gn_solve_update(Ao, bo, no, None, pose0)
    tr, rot = se3.pose_error(pose1, pose_o)
    assert tr < TIGHT and rot < TIGHT


# ------------------------------------------------------------------------------------------------- full registration
@pytest.mark.parametrize("case_name,voxel,iters", [("box_case", 0.5, 8), ("street_case", 0.8, 6)])
def test_register_matches_oracle(case_name, voxel, iters, request):
    case = request.getfixturevalue(case_name)
    om, gm = build_maps(case, 6, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(case, 6, voxel)
    mm, op = _prior(case, 6)
    o = _opts(num_iters_icp=iters, threshold_orientation_norm=1e-5)
    kps = np.zeros(len(t), dtype=cia.WPOINT3D_DTYPE)
    kps["raw_point"], kps["t"], kps["world_point"] = raw, t, world0
    frame = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
    summ = cia.CT_ICP_Registration(o).Register(gm, kps, frame, mm)
    pose_o, world_o, so = orc.register_gn(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), op, heap_mode=0)
    assert summ.success and so.success
    assert summ.num_iters == so.num_iters and summ.num_residuals_used == so.num_residuals_used
    tr, rot = se3.pose_error(frame.pose14(), pose_o)
    assert tr < POSE_TOL_M and rot < POSE_TOL_RAD
    assert tr < 1e-7 and rot < 1e-7, (tr, rot)
    assert np.abs(kps["world_point"] - world_o).max() < 1e-7
    assert abs(summ.last_step_norm - so.last_step_norm) < 1e-7
    # without the motion prior (whose location term, as written at ct_icp.cpp:892-898, pulls t_begin towards t_end at
    # driving speed) the registration moves towards the ground truth
    kps["world_point"] = world0
    frame2 = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
    cia.CT_ICP_Registration(o).Register(gm, kps, frame2, None)
    assert se3.pose_error(frame2.pose14(), sc.pose_gt)[0] < se3.pose_error(pose0, sc.pose_gt)[0]


def test_golden_vectors_through_the_gpu(golden):
    g = golden
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(float(g["resolution"]), float(g["min_dist"]),
                                                                                   int(g["max_pts"]))],
                                                default_radius=float(g["radius"])))
    kept = gm.InsertPointCloud(g["insert_points"])
    assert np.array_equal(kept, g["insert_kept"])
    s = cia.GnSolver(gm)
    s.set_debug(True)
    s.set_keypoints(g["raw"], g["world0"], g["t"])
    mm = cia.PreviousFrameMotionModel(float(g["prior_beta"][0]), float(g["prior_beta"][1]))
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(np.concatenate([[0, 0, 0, 1], g["prior_prev_b"], [0, 0, 0, 1], g["prior_prev_e"]]), 0, 0)
    o = _opts(num_iters_icp=1, min_number_neighbors=int(g["min_nb"]), max_number_neighbors=int(g["k"]),
              max_dist_to_plane_ct_icp=float(g["max_dist"]), threshold_orientation_norm=0.0)
    pose1, summ, _ = s.solve(g["pose0"], g["tbe"], o, mm)
    dbg = s.get_debug()
    A, b, n_used = s.get_system()
    assert n_used == int(g["n_used"]) and np.array_equal(dbg["n_neighbors"], g["n_neighbors"])
    assert np.array_equal(dbg["used"], g["used"])
    has = g["n_neighbors"] >= 20
    assert np.array_equal(dbg["farthest"][has], g["farthest"][has])
    assert np.allclose(A, g["A"], rtol=1e-8, atol=1e-10) and np.allclose(b, g["b"], rtol=1e-8, atol=1e-10)
    assert np.allclose(pose1, g["pose1"], atol=1e-9)
    assert np.allclose(s.world_points(), g["world1"], atol=1e-9)


# ------------------------------------------------------------------------------------------------- edge cases
def test_soft_failure_and_errors(box_case):
    om, gm = build_maps(box_case, 4, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(box_case, 4, 0.5, n_max=60)
    s = cia.GnSolver(gm)
    s.set_keypoints(raw, world0, t)
    pose1, summ, _ = s.solve(pose0, sc.t_begin_end, _opts())
    assert not summ.success and summ.num_iters == 0                      # ct_icp.cpp:860-871
    assert summ.error_log.startswith("[CT_ICP]Error : not enough keypoints selected in ct-icp !")
    assert f"Number_of_residuals : {summ.num_residuals_used}" in summ.error_log
    q0 = se3.quat_normalize(pose0[0:4])
    assert np.allclose(pose1[4:7], pose0[4:7], atol=0) and np.allclose(pose1[0:4], q0, atol=1e-15)
    assert np.array_equal(s.world_points(), world0)                      # untouched on failure at iteration 0
    _, _, so = orc.register_gn(om, raw, world0, t, pose0, sc.t_begin_end, orc.Options())
    assert so.num_residuals_used == summ.num_residuals_used
    # a timestamp outside [t_begin, t_end]: the reference CHECK-aborts (types.h:456); the ABI returns an error
    s.set_keypoints(raw, world0, t + 1.0)
    with pytest.raises(cia.CtgnError) as e:
        s.solve(pose0, sc.t_begin_end, _opts())
    assert e.value.status == L.ERR_TIMESTAMP_RANGE
    # empty keypoint set: soft failure with 0 residuals
    s.set_keypoints(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0))
    _, summ, _ = s.solve(pose0, sc.t_begin_end, _opts())
    assert not summ.success and summ.num_residuals_used == 0
    # empty map
    gm.ClearMap()
    s.set_keypoints(raw, world0, t)
    _, summ, _ = s.solve(pose0, sc.t_begin_end, _opts())
    assert not summ.success and summ.num_residuals_used == 0
    with pytest.raises(cia.CtgnError):
        s.solve(pose0, sc.t_begin_end, _opts(max_number_neighbors=33))


def test_float32_strided_keypoints(box_case):
    """ProxyView semantics (include/SlamCore/data/view.h:98-116): FLOAT32 fields with arbitrary stride are cast."""
    om, gm = build_maps(box_case, 5, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(box_case, 5, 0.6)
    rec = np.zeros(len(t), dtype=[("raw", "<f4", 3), ("junk", "<u2"), ("t", "<f8"), ("world", "<f4", 3)])
    rec["raw"], rec["t"], rec["world"] = raw, t, world0
    lib = L.lib()
    h = gm.handle
    import ctypes as C
    o = L.Options(4, 20, 20, 0, 0.3, 0.0)
    summ = L.Summary()
    pose = pose0.copy()
    tbe = np.ascontiguousarray(sc.t_begin_end)
    dp = C.POINTER(C.c_double)
    st = lib.ctgn_register(h, L.View(rec.ctypes.data + rec.dtype.fields["raw"][1], rec.strides[0], L.CTGN_F32, 0),
                           rec.ctypes.data + rec.dtype.fields["world"][1], rec.strides[0], L.CTGN_F32,
                           L.View(rec.ctypes.data + rec.dtype.fields["t"][1], rec.strides[0], L.CTGN_F64, 0), len(t),
                           pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(o), None, C.byref(summ))
    L.check(h, st)
    raw32, world32 = rec["raw"].astype(np.float64), world0.astype(np.float32).astype(np.float64)
    pose_o, world_o, so = orc.register_gn(om, raw.astype(np.float32).astype(np.float64), world32, t, pose0, sc.t_begin_end,
                                          orc.Options(4, 20, 20, False, 0.3, 0.0), None, heap_mode=0)
    assert summ.success and summ.num_residuals_used == so.num_residuals_used
    tr, rot = se3.pose_error(pose, pose_o)
    assert tr < 1e-7 and rot < 1e-7
    assert np.abs(rec["world"].astype(np.float64) - world_o).max() < 1e-5          # float32 write-back


def test_incremental_map_updates_reach_the_device(street_case):
    """Insert / evict after the map is resident: the delta upload (scatter of logged edits) must leave the device map
    identical to a map built from scratch — checked through bit-exact neighbour lists."""
    case = street_case
    om, gm = build_maps(case, 3, with_gpu=True)
    rng = np.random.default_rng(3)
    for j in range(3, 9):
        qs = case["scans"][j].world_gt[rng.choice(len(case["scans"][j].world_gt), 200, replace=False)]
        for q, g in zip(qs, gm.ComputeNeighborhoods(qs, 20)):            # forces residency / sync before the edits
            assert np.array_equal(g, om.radius_search(q, 0.0, 20, heap_mode=0))
        pts = case["scans"][j].world_gt
        gm.InsertPointCloud(pts)
        om.insert(pts)
        loc = case["scans"][j].pose_gt[11:14]
        gm.RemoveElementsFarFromLocation(loc, 30.0)
        om.remove_far(loc, 30.0)
        assert gm.NumPoints() == om.num_points()
    qs = case["scans"][9].world_gt[rng.choice(len(case["scans"][9].world_gt), 500, replace=False)]
    for q, g in zip(qs, gm.ComputeNeighborhoods(qs, 20)):
        assert np.array_equal(g, om.radius_search(q, 0.0, 20, heap_mode=0))


def test_stepwise_api_equals_fused_loop(box_case):
    """The three spellings of the GN loop: stepwise entry points (gn_accumulate / gn_solve_update), the fused three-launch loop, and —
    round 3, opt-in for small frames — the ONE-launch persistent kernel (ctgn_set_persistent). The first two give identical bits
    (same kernels, deterministic reductions); the persistent kernel sums the same per-keypoint terms in its own fixed order (per wave,
    per block, blocks in index order): identical discrete results, pose / world points / system equal to 1e-12, and bit-identical to
    itself from run to run."""
    om, gm = build_maps(box_case, 5, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(box_case, 5, 0.5)
    o = _opts(num_iters_icp=4, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)

    def stepwise(r, w, tt, prior=None):
        s.set_keypoints(r, w, tt)
        s.gn_begin(pose0, sc.t_begin_end, o, prior)
        for _ in range(4):
            s.gn_accumulate()
            s.gn_solve_update()
        done = s.gn_done()
        pose, summ, _ = s.gn_end()
        return pose, summ, s.world_points(), s.get_system(), done

    def fused(r, w, tt, prior=None, persistent=0):
        s.set_persistent(persistent)
        s.set_keypoints(r, w, tt)
        pose, summ, _ = s.solve(pose0, sc.t_begin_end, o, prior)
        out = pose, summ, s.world_points(), s.get_system()
        s.set_persistent(0)
        return out

    pose_s, summ_s, w_s, sys_s, done = stepwise(raw, world0, t)
    assert not done and summ_s.num_iters == 4
    # small frames take 64-thread residual blocks (more CUs for the scattered gathers), large ones 256-thread blocks; odd block counts,
    # ragged last groups, soft failure (< 100 used)
    for n in (5000, 2048, 2047, 1300, 1025, 1024, 777, 256, 65, 64, 40):
        idx = np.arange(n) % len(t)
        for prior in (None, _prior(box_case, 5)[0]):
            pose_s, summ_s, w_s, sys_s, _ = stepwise(raw[idx], world0[idx], t[idx], prior)
            pose_f, summ_f, w_f, sys_f = fused(raw[idx], world0[idx], t[idx], prior, persistent=0)
            assert np.array_equal(pose_f, pose_s) and np.array_equal(w_f, w_s), n
            assert summ_s.success == summ_f.success and summ_s.num_iters == summ_f.num_iters and summ_s.num_residuals_used == summ_f.num_residuals_used
            assert np.array_equal(sys_f[0], sys_s[0]) and np.array_equal(sys_f[1], sys_s[1]) and sys_f[2] == sys_s[2]
            pose_p, summ_p, w_p, sys_p = fused(raw[idx], world0[idx], t[idx], prior, persistent=1)            # one persistent launch when n <= 1024
            assert summ_p.success == summ_s.success and summ_p.num_iters == summ_s.num_iters and summ_p.num_residuals_used == summ_s.num_residuals_used, n
            assert np.abs(pose_p - pose_s).max() < 1e-12 and np.abs(w_p - w_s).max() < 1e-11, n
            assert sys_p[2] == sys_s[2] and np.abs(sys_p[0] - sys_s[0]).max() <= 1e-12 * max(np.abs(sys_s[0]).max(), 1e-300)
            pose_q, summ_q, w_q, sys_q = fused(raw[idx], world0[idx], t[idx], prior, persistent=1)
            assert np.array_equal(pose_p, pose_q) and np.array_equal(w_p, w_q) and np.array_equal(sys_p[0], sys_q[0])


def test_pools_give_the_same_neighbours_as_searching_every_iteration(box_case, nclt_case):
    """Round 3: neighbour pools (ctgn_set_pools). With pools a later iteration first checks the pool its previous
    bounded search left (gathered pool members + one selection, conservative certificate) and searches only what that does not cover.
    The neighbour sets and their order must be the ones a search finds: pose, world points, packed system and the captured per-keypoint
    neighbourhoods (count, normal, a2D, farthest neighbour) are bit-identical with pools on and off, after every iteration — on the
    27-voxel sweep (box), the 125-voxel sweep with sparse neighbourhoods in use (NCLT profile: min 10 < max 20), with the whole pose
    error moving the keypoints (large perturbation: few pools certify) and with a converged pose (all of them do), and through the
    robust route. The instrumented instantiation counts what the pools certified: nothing in iterations 0 / 1, most keypoints later."""
    cases = []
    om, gm = build_maps(box_case, 5, with_gpu=True)
    for perturb in ((0.005, 0.03), (0.0, 0.0), (0.02, 0.2)):
        sc, raw, t, pose0, world0 = _keypoints(box_case, 5, 0.3, perturb=perturb)
        cases.append((gm, sc, raw, t, pose0, world0, _opts(num_iters_icp=6, threshold_orientation_norm=0.0), _prior(box_case, 5)[0]))
    omn, gmn = build_maps(nclt_case, 8, with_gpu=True)
    scn = nclt_case["scans"][8]
    sel = syn.grid_sample_indices(scn.raw, 0.8)[:1500]
    pose_n = syn.perturb_pose(scn.pose_gt, 0.01, 0.05, seed=7)
    cases.append((gmn, scn, scn.raw[sel], scn.t[sel], pose_n, se3.ct_transform(pose_n, scn.t_begin_end, scn.t[sel], scn.raw[sel]),
                  _opts(num_iters_icp=8, min_number_neighbors=10, threshold_orientation_norm=0.0), _prior(nclt_case, 8)[0]))
    for gmap, sc, raw, t, pose0, world0, o, prior in cases:
        runs = []
        # pools off | pools on, phase V inside the search kernel | pools on with the split launches forced (round 4: from the third search on
        # the check is a kernel of its own, k_pool_check, and the search kernel runs over the list of positions it could not certify;
        # by default only from 400 k keypoints)
        # ... | pools on, the searches whose carried-over bound lies beyond the radius keep none (bit 29: as before round 5)
        for pools, mask in ((0, 0), (1, 0), (1, 1 << 25), (1, 1 << 29)):
            s = cia.GnSolver(gmap)
            s.set_pools(pools)
            s.set_ablation(mask)
            s.set_debug(True)
            s.set_keypoints(raw, world0, t)
            s.gn_begin(pose0, sc.t_begin_end, o, prior)
            per_iter = []
            for _ in range(o.num_iters_icp):
                s.gn_iterate(1)
                d = s.get_debug()
                per_iter.append((s.get_system(), d["n_neighbors"].copy(), d["normal"].copy(), d["a2d"].copy(), d["farthest"].copy(), d["used"].copy()))
            pose, summ, _ = s.gn_end()
            runs.append((pose, summ, s.world_points(), per_iter))
        pose_a, summ_a, w_a, it_a = runs[0]
        for which, (pose_b, summ_b, w_b, it_b) in enumerate(runs[1:]):
            assert summ_a.success and summ_a.num_iters == summ_b.num_iters == o.num_iters_icp and summ_a.num_residuals_used == summ_b.num_residuals_used
            assert np.array_equal(pose_a, pose_b) and np.array_equal(w_a, w_b), which
            for k_it, (a, b) in enumerate(zip(it_a, it_b)):
                assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1]) and a[0][2] == b[0][2], (which, k_it)
                for x, y in zip(a[1:], b[1:]):
                    assert np.array_equal(x, y), (which, k_it)
        # the instrumented instantiation says how many keypoints the pools certified per iteration
        s = cia.GnSolver(gmap)
        s.set_pools(1); s.set_variant(3)
        s.set_keypoints(raw, world0, t)
        s.gn_begin(pose0, sc.t_begin_end, o, prior)
        s.phase_cycles(reset=True)
        certified = []
        for _ in range(o.num_iters_icp):
            s.gn_iterate(1)
            certified.append(int(s.phase_cycles(reset=True)[8]))
        pose_v, _, _ = s.gn_end()
        assert np.array_equal(pose_v, pose_a)
        assert certified[0] == 0 and certified[1] == 0 and certified[-1] > 0.5 * len(t), certified
    # the robust route searches once per outer iteration through the same kernel
    sc, raw, t, pose0, world0 = _keypoints(box_case, 5, 0.3)
    o = cia.CTICPOptions(solver=cia.CERES, debug_print=False, num_iters_icp=5, ls_max_num_iters=3)
    if o is not None:
        outs = []
        for pools in (0, 1):
            s = cia.GnSolver(gm)
            s.set_pools(pools)
            s.set_keypoints(raw, world0, t)
            pose_r, summ_r, _ = s.solve_robust(pose0, sc.t_begin_end, o)
            outs.append((pose_r, s.world_points(), s.robust_blocks()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert np.array_equal(outs[0][2]["rank"], outs[1][2]["rank"])
        kept = outs[0][2]["rank"] >= 0                      # blocks the route built (the arrays of a keypoint without a block are never written)
        assert kept.sum() > 100
        for key in outs[0][2]:
            assert np.array_equal(np.asarray(outs[0][2][key])[kept], np.asarray(outs[1][2][key])[kept]), key


def test_split_pool_check_counters_survive_solves_that_stop_early(box_case):
    """Round 5 (advisor): the split pool-check launches count failing positions in two alternating counters; a launch behind the stop test
    returns before it zeroes the next one while the host keeps toggling the slot, so after a solve that converged early a stale count could
    sit in the slot the next solve starts in. Each solve now starts from zeroed counters and slot 0. Here: on ONE handle, fused solves that stop
    after a varying number of their enqueued launches (both parities of skipped split launches), each followed by a solve on FEWER keypoints;
    every result must be the one a fresh handle without pools gives."""
    om, gm = build_maps(box_case, 5, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(box_case, 5, 0.3, perturb=(0.005, 0.03))
    prior = _prior(box_case, 5)[0]

    def fresh(n, o):
        s = cia.GnSolver(gm)
        s.set_pools(0)
        s.set_keypoints(raw[:n], world0[:n], t[:n])
        pose, summ, _ = s.solve(pose0, sc.t_begin_end, o, prior)
        return pose, summ, s.world_points()

    shared = cia.GnSolver(gm)
    shared.set_pools(1)
    shared.set_ablation(1 << 25)
    stops = set()
    for thr in (3e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2):                                                     # a threshold the solve meets after 3 .. 10 iterations
        stop_at = fresh(len(t), _opts(num_iters_icp=14, threshold_orientation_norm=thr))[1].num_iters  # where this solve's stop test fires
        if 3 <= stop_at <= 10:
            break
    assert 3 <= stop_at <= 10, stop_at                                                                   # late enough for split launches to have run
    for num_iters in (stop_at + 1, stop_at + 2, stop_at + 3, stop_at + 4):                               # 1 .. 4 launches skipped behind the stop
        o = _opts(num_iters_icp=num_iters, threshold_orientation_norm=thr)
        for n in (len(t), len(t) // 3):
            shared.set_keypoints(raw[:n], world0[:n], t[:n])
            pose_s, summ_s, _ = shared.solve(pose0, sc.t_begin_end, o, prior)
            pose_f, summ_f, w_f = fresh(n, o)
            assert summ_s.success == summ_f.success and summ_s.num_iters == summ_f.num_iters and summ_s.num_residuals_used == summ_f.num_residuals_used
            assert np.array_equal(pose_s, pose_f) and np.array_equal(shared.world_points(), w_f), (num_iters, thr, n)
            if summ_s.num_iters >= 3 and summ_s.num_iters < num_iters:
                stops.add((num_iters - summ_s.num_iters) % 2)
    assert stops == {0, 1}, stops           # both parities of launches skipped behind the stop test were exercised


# ------------------------------------------------------------------------------------------------- full-size properties
@pytest.fixture(scope="module")
def config_b_full():
    """BASELINE.json configs[1] at full size: HDL-64E sweep (~130 k returns) over the procedural street, driving
    profile map (0.8 m x 30 pts, radius 0.75), 10 map frames."""
    scene = syn.street_scene(300.0, seed=1)
    dirs, rel_t = syn.lidar_pattern("hdl64")
    knots = syn.driving_trajectory(12, seed=0, start_x=20.0)
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75))
    om = orc.Map(resolutions=[(0.8, 0.1, 30)], default_radius=0.75)
    for j in range(10):
        sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), noise=0.02, seed=100 + j)
        pts = sc.world_gt[syn.grid_sample_indices(sc.raw, 0.5)]
        assert np.array_equal(gm.InsertPointCloud(pts), om.insert(pts))
    sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, 10), 1.0, 1.1, noise=0.02, seed=110)
    _B2_ORACLE_MAP["om"] = om
    return gm, sc


_B2_ORACLE_MAP = {}


def test_full_size_properties(config_b_full):
    gm, sc = config_b_full
    n = len(sc.t)
    assert n > 100_000
    pose0 = syn.perturb_pose(sc.pose_gt, 0.003, 0.03, seed=4)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, sc.t, sc.raw)
    o = _opts(num_iters_icp=1, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    results = {}
    for variant in (0, 1, 2, 5):
        s.set_variant(variant)
        s.set_keypoints(sc.raw, world0, sc.t)
        pose1, summ, _ = s.solve(pose0, sc.t_begin_end, o)
        results[variant] = (s.get_system(), pose1, summ.num_residuals_used)
    (A0, b0, n0), p0, _ = results[0]
    assert n0 > 20_000
    for v in (1, 2, 5):                                # independent search kernels / paths agree at full size
        (A, b, nu), p, _ = results[v]
        assert nu == n0
        assert np.abs(A - A0).max() < 1e-9 * np.abs(A0).max() and np.abs(b - b0).max() < 1e-9 * np.abs(b0).max() + 1e-13
        assert se3.pose_error(p, p0)[0] < 1e-9
    # permutation invariance: the sum over keypoints does not depend on their order (up to FP64 rounding)
    perm = np.random.default_rng(0).permutation(n)
    s.set_variant(0)
    s.set_keypoints(sc.raw[perm], world0[perm], sc.t[perm])
    pose_p, summ_p, _ = s.solve(pose0, sc.t_begin_end, o)
    Ap, bp, npn = s.get_system()
    assert npn == n0 and np.abs(Ap - A0).max() < 1e-9 * np.abs(A0).max()
    wp = s.world_points()
    # re-transform property: the returned world points are T(pose) raw for the returned pose
    assert np.abs(wp - se3.ct_transform(pose_p, sc.t_begin_end, sc.t[perm], sc.raw[perm])).max() < 1e-9
    # linearity of the packed system: accumulating two halves separately and adding equals the whole
    half = n // 2
    parts = []
    for sl in (slice(0, half), slice(half, n)):
        s.set_keypoints(sc.raw[sl], world0[sl], sc.t[sl])
        s.solve(pose0, sc.t_begin_end, o)
        parts.append(s.get_system())
    assert parts[0][2] + parts[1][2] == n0
    assert np.abs(parts[0][0] + parts[1][0] - A0).max() < 1e-9 * np.abs(A0).max()
    # traffic counters are exactly 27 probes per in-range keypoint
    s.set_keypoints(sc.raw, world0, sc.t)
    probed, hit, pts = s.count_traffic()
    assert probed == 27 * n and 0 < hit < probed and pts > 20 * n


def test_cpp_adapter_program():
    """The C++ host adapter (ct_icp_amd/cpp/ct_icp_gpu.hpp: the reference's class names over the C ABI) registers a scan
    and recovers the known rigid offset."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ct_icp_amd", "cpp", "adapter_check")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("adapter ok") and "adapter-ceres ok" in out.stdout, out.stdout + out.stderr
    assert "adapter-frame ok" in out.stdout, out.stdout                      # RegisterFrame / UpdateMapFromFrame (frame pipeline)


def test_sharded_loop_single_rank_equals_fused(box_case):
    """The multi-GPU loop (accumulate -> all-reduce of the 96-double system over RCCL -> solve, ct_icp_amd/distributed.py)
    on one rank gives bit-identical results to the fused loop: exercises the external stream / external system buffer."""
    import os
    import torch
    import torch.distributed as dist
    from ct_icp_amd.distributed import ShardedGnSolver
    om, gm = build_maps(box_case, 5, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(box_case, 5, 0.5)
    o = _opts(num_iters_icp=4, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    s.set_keypoints(raw, world0, t)
    pose_f, summ_f, _ = s.solve(pose0, sc.t_begin_end, o)
    w_f = s.world_points()
    created = False
    if not dist.is_initialized():
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
        created = True
    try:
        # both flavours: the collective issued by the library (ncclAllReduce from C, ctgn_solve_sharded) and the stepwise loop with
        # torch.distributed.all_reduce in between
        for library_collective in (True, False):
            sh = ShardedGnSolver(gm, library_collective=library_collective)
            sh.set_keypoints(raw, world0, t)
            pose_s, summ_s, _ = sh.solve(pose0, sc.t_begin_end, o)
            torch.cuda.synchronize()
            assert np.array_equal(pose_f, pose_s) and summ_s.num_iters == 4
            assert np.array_equal(w_f, sh.solver.world_points())
            A, b, n = sh.solver.get_system()
            assert n == summ_s.num_residuals_used
            # a rank whose shard cannot start (timestamp outside the frame) still takes part in every exchange with a poisoned count,
            # so the job fails instead of hanging its peers in the all-reduce; the handle stays usable
            t_bad = t.copy()
            t_bad[3] = sc.t_begin_end[1] + 1.0
            sh.set_keypoints(raw, world0, t_bad)
            with pytest.raises(Exception, match="timestamp|TIMESTAMP"):
                sh.solve(pose0, sc.t_begin_end, o)
            sh.set_keypoints(raw, world0, t)
            pose_s2, _, _ = sh.solve(pose0, sc.t_begin_end, o)
            assert np.array_equal(pose_f, pose_s2)
            sh.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_wave_shared_probes_of_the_125_voxel_sweep(nclt_case):
    """Round 4: on the 125-voxel sweep a round whose (active) keypoints live in one home voxel can probe its 125 voxels once per wave
    (rows_tiles, shared2) instead of eight dependent batches per row. Dense keypoints (every return of the scan) worked through in
    home-voxel order make that the rule; with pools on and off, four iterations: counts, farthest neighbours, gates and packed system
    of the per-row probes and of the shared ones (bit 22 of the ablation mask switches them ON: they are off by default, see
    CTGN_SHARED2_DEFAULT_ON in ctgn_kernels.hpp) are equal after every iteration, and the first accumulation's neighbour sets are the
    oracle's. Round 5: the same against the batch loop run to its end (bit 26: no early stop after the batches some row can reach),
    against a shared probe table that is never reused (bit 27), and with tiles of four consecutive rounds (tuning tile_chunk:
    consecutive rounds share their home voxel, so the table IS reused). Round 6: the same with the home-voxel-group stage (tuning
    stage_lds: the candidates of a group streamed from LDS, rows_tiles STAGE), alone, with chunked tiles, and over the shared probes."""
    S2 = 1 << 22
    case = nclt_case
    om, gm = build_maps(case, 8, with_gpu=True)
    sc = case["scans"][8]
    raw, t = sc.raw[:12000], sc.t[:12000]
    pose0 = syn.perturb_pose(sc.pose_gt, 0.01, 0.05, seed=9)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, t, raw)
    o = _opts(num_iters_icp=4, min_number_neighbors=10, threshold_orientation_norm=0.0)
    for pools in (0, 1):
        runs = []
        for mask, chunk, stage in ((0, 0, 0), (S2, 0, 0), (1 << 26, 0, 0), (S2 | (1 << 26), 0, 0), (S2 | (1 << 27), 0, 0), (S2, 4, 0), (0, 4, 0),
                                   (0, 0, 1), (0, 4, 1), (S2, 16, 1)):
            L.lib().ctgn_set_tuning(b"tile_chunk", float(chunk))
            L.lib().ctgn_set_tuning(b"stage_lds", float(stage))
            try:
                s = cia.GnSolver(gm)
                s.phase_cycles(reset=True)
                s.set_ordering(1)
                s.set_pools(pools)
                s.set_ablation(mask)
                s.set_debug(True)
                s.set_keypoints(raw, world0, t)
                s.gn_begin(pose0, sc.t_begin_end, o, None)
                per_iter = []
                for _ in range(o.num_iters_icp):
                    s.gn_iterate(1)
                    d = s.get_debug()
                    per_iter.append((s.get_system(), d["n_neighbors"].copy(), d["farthest"].copy(), d["used"].copy(), d["a2d"].copy()))
                pose, summ, _ = s.gn_end()
                if stage:
                    # the prototype's path did run: a good part of the search rounds streamed a staged table (the others reach beyond the
                    # inner voxels, mix home voxels, or their group holds more points than the table)
                    fills, staged_rounds, all_rounds = (int(v) for v in s.phase_cycles()[7:10])
                    assert all_rounds > 1000 and staged_rounds > 0.25 * all_rounds and 0 < fills <= staged_rounds, (fills, staged_rounds, all_rounds)
                    if chunk:
                        assert fills < staged_rounds, (fills, staged_rounds)       # consecutive rounds of a chunk share a table
            finally:
                L.lib().ctgn_set_tuning(b"tile_chunk", 0.0)
                L.lib().ctgn_set_tuning(b"stage_lds", 0.0)
            runs.append((pose, per_iter))
        for other in runs[1:]:
            assert np.array_equal(runs[0][0], other[0])
            for k_it, (a, b) in enumerate(zip(runs[0][1], other[1])):
                assert np.array_equal(a[0][0], b[0][0]) and a[0][2] == b[0][2], k_it
                for x, y in zip(a[1:], b[1:]):
                    assert np.array_equal(x, y), k_it
    _, _, _, info = orc.gn_accumulate(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), heap_mode=0, debug=True)
    first = runs[0][1][0]
    assert np.array_equal(first[1], info["n_neighbors"])
    has = info["n_neighbors"] >= 10
    assert has.sum() > 2000 and np.array_equal(first[2][has], info["farthest"][has]) and np.array_equal(first[3], info["used"])


def test_config_c_nclt_profile_matches_oracle(nclt_case):
    case = nclt_case
    om, gm = build_maps(case, 8, with_gpu=True)
    assert gm.SearchParamsFromRadiusSearch() == (0, 0.5, 2) == om.search_params()
    sc = case["scans"][8]
    sel = syn.grid_sample_indices(sc.raw, 0.8)[:1500]                 # max_num_keypoints 1500
    raw, t = sc.raw[sel], sc.t[sel]
    pose0 = syn.perturb_pose(sc.pose_gt, 0.01, 0.05, seed=7)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, t, raw)
    o = _opts(num_iters_icp=20, min_number_neighbors=10, threshold_orientation_norm=1e-4)
    mm, op = _prior(case, 8)
    s = cia.GnSolver(gm)
    s.set_debug(True)
    s.set_keypoints(raw, world0, t)
    pose1, summ, _ = s.solve(pose0, sc.t_begin_end, o, mm)
    pose_o, world_o, so = orc.register_gn(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), op, heap_mode=0)
    assert summ.success and so.success and summ.num_iters == so.num_iters
    assert summ.num_residuals_used == so.num_residuals_used > 300
    tr, rot = se3.pose_error(pose1, pose_o)
    assert tr < 1e-7 and rot < 1e-7, (tr, rot)
    assert np.abs(s.world_points() - world_o).max() < 1e-7
    # neighbourhoods with 10..19 points are used under this profile (min 10 < max 20)
    nn = s.get_debug()["n_neighbors"]
    assert ((nn >= 10) & (nn < 20)).sum() > 0
    # all three resolutions of the host mirror agree with the oracle map
    for li in range(3):
        assert gm.NumVoxels(li) == om.num_voxels(li)
    # the 125-voxel instantiation in home-voxel order (position-ordered working copy, XCD-split hand-out at 94 blocks), with and
    # without debug capture (the latter goes through the order indirection): same registration
    for debug in (False, True):
        s.set_ordering(1)
        s.set_debug(debug)
        s.set_keypoints(raw, world0, t)
        pose2, summ2, _ = s.solve(pose0, sc.t_begin_end, o, mm)
        assert summ2.num_iters == summ.num_iters and summ2.num_residuals_used == summ.num_residuals_used
        assert np.abs(pose2 - pose1).max() < 1e-11 and np.abs(s.world_points() - world_o).max() < 1e-7
    assert np.array_equal(s.get_debug()["n_neighbors"], nn)


def test_home_voxel_ordering_changes_nothing_but_the_schedule(config_b_full, box_case):
    """ctgn_set_ordering (include/ctgn.h): working through an upload in home-voxel order (positions sorted on the device, the
    kernels iterate on a position-ordered working copy, XCD-split tile hand-out) is a scheduling decision. Per-keypoint
    results — neighbour counts, normals, planarity, used flags — must be bit-identical (debug capture keeps the caller-order
    arrays and goes through the order indirection), the packed system equal up to summation order, poses and world points
    equal to rounding, on a full scan (132 k keypoints) and on a small frame; the automatic mode orders from 13 iterations on
    at this size and must give the forced result bit for bit."""
    gm, sc = config_b_full
    pose0 = syn.perturb_pose(sc.pose_gt, 0.003, 0.03, seed=4)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, sc.t, sc.raw)
    s = cia.GnSolver(gm)
    out = {}
    for mode in (0, 1):
        s.set_ordering(mode)
        s.set_debug(True)
        s.set_keypoints(sc.raw, world0, sc.t)
        s.solve(pose0, sc.t_begin_end, _opts(num_iters_icp=1, threshold_orientation_norm=0.0))
        dbg = s.get_debug()
        s.set_debug(False)
        s.set_keypoints(sc.raw, world0, sc.t)
        pose, summ, _ = s.solve(pose0, sc.t_begin_end, _opts(num_iters_icp=4, threshold_orientation_norm=0.0))
        out[mode] = (dbg, pose, summ, s.world_points(), s.get_system())
    d0, p0, s0, w0, (A0, b0, n0) = out[0]
    d1, p1, s1, w1, (A1, b1, n1) = out[1]
    for key in d0:
        assert np.array_equal(d0[key], d1[key]), key
    assert s0.num_residuals_used == s1.num_residuals_used and n0 == n1 and s0.num_iters == s1.num_iters == 4
    assert np.abs(A1 - A0).max() < 1e-12 * np.abs(A0).max() and np.abs(b1 - b0).max() < 1e-12 * np.abs(b0).max() + 1e-16
    assert np.abs(p1 - p0).max() < 1e-12 and np.abs(w1 - w0).max() < 1e-10
    s.set_ordering(-1)                                  # automatic: 16 iterations of 132 k keypoints cover the sort
    s.set_keypoints(sc.raw, world0, sc.t)
    pa, _, _ = s.solve(pose0, sc.t_begin_end, _opts(num_iters_icp=16, threshold_orientation_norm=0.0))
    s.set_ordering(1)
    s.set_keypoints(sc.raw, world0, sc.t)
    pf, _, _ = s.solve(pose0, sc.t_begin_end, _opts(num_iters_icp=16, threshold_orientation_norm=0.0))
    s.set_ordering(0)
    s.set_keypoints(sc.raw, world0, sc.t)
    pn, _, _ = s.solve(pose0, sc.t_begin_end, _opts(num_iters_icp=16, threshold_orientation_norm=0.0))
    assert np.array_equal(pa, pf) and not np.array_equal(pa, pn) and np.abs(pa - pn).max() < 1e-11
    # a small frame (a few hundred keypoints, one block): forced ordering against the oracle
    om, gb = build_maps(box_case, 4, with_gpu=True)
    scb, raw, t, poseb, worldb = _keypoints(box_case, 4, 0.25)
    sb = cia.GnSolver(gb)
    sb.set_ordering(1)
    sb.set_keypoints(raw, worldb, t)
    pg, sg, _ = sb.solve(poseb, scb.t_begin_end, _opts())
    po, _, so = orc.register_gn(om, raw, worldb.copy(), t, poseb, scb.t_begin_end, _oopts(_opts()), None, heap_mode=0)
    assert sg.success and sg.num_residuals_used == so.num_residuals_used and sg.num_iters == so.num_iters
    tr, rot = se3.pose_error(pg, po)
    assert tr < 1e-7 and rot < 1e-7


def test_per_xcd_presums_change_nothing_but_the_summation_order(config_b_full):
    """Round 5, ctgn_kernels.hpp XcdReduce: on a full sweep the residual kernel's per-block records (518 of them) are pre-summed by the
    last block of each of 32 groups (four per XCD, each out of its own L2) and the solve kernel adds 32 group records: same keypoints
    used, system equal up to summation order, poses to rounding — and the path must actually have run (ctgn_path_counters), with the
    placement check (per-XCC arrival counts inside the groups' 64-bit tickets) holding on this part; the pad entries of the packed system
    stay zero."""
    gm, sc = config_b_full
    pose0 = syn.perturb_pose(sc.pose_gt, 0.003, 0.03, seed=4)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, sc.t, sc.raw)
    s = cia.GnSolver(gm)
    out = {}
    try:
        for on in (1, 0):
            L.lib().ctgn_set_tuning(b"xcd_reduce", float(on))
            c0 = s.path_counters()
            s.set_keypoints(sc.raw, world0, sc.t)
            pose, summ, _ = s.solve(pose0, sc.t_begin_end, _opts(num_iters_icp=4, threshold_orientation_norm=0.0))
            c1 = s.path_counters()
            raw96 = np.zeros(96)
            L.check(s._h, L.lib().ctgn_get_system(s._h, raw96.ctypes.data_as(C.POINTER(C.c_double))))
            out[on] = (pose, summ, s.get_system(), c1[0] - c0[0], c1[1], raw96)
    finally:
        L.lib().ctgn_set_tuning(b"xcd_reduce", -1.0)
    (p1, s1, (A1, b1, n1), launches1, groups1, r1), (p0, s0, (A0, b0, n0), launches0, groups0, r0) = out[1], out[0]
    assert launches1 == 4 and groups1 == 1, (launches1, groups1)        # four residual launches pre-summed, the last solve summed group records
    assert launches0 == 0                                               # (the solve kernel's path word is written only when pre-sums are on)
    assert np.all(r1[91:] == 0.0) and np.all(r0[91:] == 0.0)
    assert s1.num_residuals_used == s0.num_residuals_used and n1 == n0 and s1.num_iters == s0.num_iters == 4
    assert np.abs(A1 - A0).max() < 1e-12 * np.abs(A0).max() and np.abs(b1 - b0).max() < 1e-12 * np.abs(b0).max() + 1e-16
    assert np.abs(p1 - p0).max() < 1e-12


def test_full_scan_undistortion(config_b_full):
    """the continuous-time transform of a whole sweep (reference src/ct_icp/odometry.cpp:461-486) against
    the oracle's InterpolatePose * raw, on ~130 k points."""
    gm, sc = config_b_full
    pose = syn.perturb_pose(sc.pose_gt, 0.01, 0.05, seed=9)
    got = cia.transform_points(gm, sc.raw, sc.t, pose, sc.t_begin_end)
    idx = np.random.default_rng(0).choice(len(sc.t), 3000, replace=False)
    want = orc.transform_points(pose, sc.t_begin_end, sc.t[idx], sc.raw[idx])
    assert np.abs(got[idx] - want).max() < 1e-11
    assert np.abs(got - se3.ct_transform(pose, sc.t_begin_end, sc.t, sc.raw)).max() < 1e-10
    with pytest.raises(cia.CtgnError) as e:
        cia.transform_points(gm, sc.raw[:10], sc.t[:10] + 5.0, pose, sc.t_begin_end)
    assert e.value.status == L.ERR_TIMESTAMP_RANGE


def _sorted_rows(a):
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def test_device_resident_map_maintenance(street_case, box_case):
    """InsertPointInVoxelMap (map.h:261-293) and RemoveElementsFarFromLocation (:305-322) executed ON the
    GPU (stable sort by voxel key + one thread per voxel in original order) give the oracle's insert decisions, point sets,
    evictions, neighbour lists and registration — through several table growths, tombstones and block reuse."""
    case = street_case
    res = [(0.4, 0.05, 20), (0.8, 0.1, 30), (1.6, 0.15, 40)]
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=0.9,
                                                device_updates=True))
    om = orc.Map(resolutions=res, default_radius=0.9)
    rng = np.random.default_rng(1)
    for j in range(8):
        sc = case["scans"][j]
        pts = sc.world_gt[rng.permutation(len(sc.world_gt))[:20000]]
        kg, ko = gm.InsertPointCloud(pts), om.insert(pts)
        assert np.array_equal(kg, ko), f"insert decisions differ in frame {j}"
        assert gm.NumPoints() == om.num_points()
        if j >= 2:
            loc = sc.pose_gt[11:14]
            gm.RemoveElementsFarFromLocation(loc, 35.0)
            om.remove_far(loc, 35.0)
            assert gm.NumPoints() == om.num_points()
        for li in range(3):
            assert gm.NumVoxels(li) == om.num_voxels(li)
    for li in range(3):
        assert np.array_equal(_sorted_rows(gm.MapAsPointCloud(li)), _sorted_rows(om.export(li)))
    assert gm.SearchParamsFromRadiusSearch() == om.search_params() == (1, 0.8, 2)
    qs = case["scans"][8].world_gt[rng.choice(len(case["scans"][8].world_gt), 400, replace=False)]
    for q, g in zip(qs, gm.ComputeNeighborhoods(qs, 20)):
        assert np.array_equal(g, om.radius_search(q, 0.0, 20, heap_mode=0))
    # float32 strided input goes through the same cast as the host path
    rec = np.zeros(3000, dtype=[("pad", "<f4"), ("xyz", "<f4", 3)])
    rec["xyz"] = case["scans"][9].world_gt[:3000]
    assert np.array_equal(gm.InsertPointCloud(rec["xyz"]), om.insert(rec["xyz"].astype(np.float64)))
    # a registration on the device-maintained map equals the oracle's
    sc, raw, t, pose0, world0 = _keypoints(case, 9, 0.8)
    o = _opts(num_iters_icp=4, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    s.set_keypoints(raw, world0, t)
    pose1, summ, _ = s.solve(pose0, sc.t_begin_end, o)
    pose_o, _, so = orc.register_gn(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), None, heap_mode=0)
    assert summ.num_residuals_used == so.num_residuals_used and se3.pose_error(pose1, pose_o)[0] < 1e-7
    gm.ClearMap()
    assert gm.NumPoints() == 0 and len(gm.MapAsPointCloud(1)) == 0
    gm.InsertPointCloud(box_case["scans"][0].world_gt)
    om2 = orc.Map(resolutions=res, default_radius=0.9)
    om2.insert(box_case["scans"][0].world_gt)
    assert gm.NumPoints() == om2.num_points()


def test_grid_sampling_on_device(config_b_full):
    """sub_sample_frame (reference src/ct_icp/ct_icp.cpp:65-83) on the GPU keeps exactly the points the
    oracle keeps — the first point of every voxel — for the reference's two stages (0.5 m frame grid, 1.5 m keypoint grid)."""
    gm, sc = config_b_full
    for size in (0.5, 1.5, 0.05):
        got = cia.grid_sampling(gm, sc.raw, size)
        want = orc.grid_sampling(sc.raw, size)
        assert len(got) == len(want) <= len(sc.raw)                       # test_A_grid_sampling.cxx:7-23
        assert np.array_equal(got, want)                                  # same set AND first-insertion order
    sub = sc.raw[cia.grid_sampling(gm, sc.raw, 0.5)]
    kp = cia.grid_sampling(gm, sub, 1.5)
    assert np.array_equal(np.sort(kp), np.sort(orc.grid_sampling(sub, 1.5))) and 500 < len(kp) < 5000
    assert len(cia.grid_sampling(gm, np.zeros((0, 3)), 1.0)) == 0


def test_adaptive_sampling_on_device(config_b_full):
    """SURVEY 8f row 2, adaptive variant: AdaptiveSamplePointsInGrid (reference include/ct_icp/algorithm/sampling.h:55-110, the
    NCLT profile's keypoint sampling) on the GPU returns exactly the oracle's indices in the same order, for the default bands,
    several points per voxel, the max_num_points stop (max + 1 survive), custom bands, float32 strided input and a CUDA tensor."""
    import torch
    gm, sc = config_b_full
    raw = sc.raw
    for k, mx in ((1, -1), (3, -1), (1, 1500), (2, 7)):
        o = cia.AdaptiveGridSamplingOptions(num_points_per_voxel=k, max_num_points=mx)
        got = cia.AdaptiveSamplePointsInGrid(gm, raw, o)
        want = orc.adaptive_sampling(raw, o.distance_voxel_size, k, mx)
        assert np.array_equal(got, want) and (mx < 0 or len(got) == mx + 1)
    d = np.linalg.norm(raw[cia.AdaptiveSamplePointsInGrid(gm, raw)], axis=1)
    assert d.min() >= 0.5 and d.max() < 200.0
    custom = cia.AdaptiveGridSamplingOptions(distance_voxel_size=[(1.0, 0.5), (30.0, 2.0), (60.0, -1.0)], num_points_per_voxel=2)
    assert np.array_equal(cia.AdaptiveSamplePointsInGrid(gm, raw, custom), orc.adaptive_sampling(raw, custom.distance_voxel_size, 2, -1))
    raw_d = torch.from_numpy(raw).to("cuda:0")
    got_d = cia.AdaptiveSamplePointsInGrid(gm, raw_d)
    assert got_d.is_cuda and np.array_equal(got_d.cpu().numpy().astype(np.uint32), orc.adaptive_sampling(raw))
    on_first = np.array([[0.5, 0.0, 0.0], [0.3, 0.4, 0.0], [1.0, 0.0, 0.0], [0.1, 0.0, 0.0], [0.0, 250.0, 0.0]])
    assert cia.AdaptiveSamplePointsInGrid(gm, on_first).tolist() == [2]
    assert len(cia.AdaptiveSamplePointsInGrid(gm, np.zeros((0, 3)))) == 0
    for bad in ([(2.0, 0.1), (1.0, 0.2)], [(1.0, 0.1)], [(0.5, -1.0), (2.0, 0.1)], [(0.5, 1e-6), (200.0, -1.0)]):
        with pytest.raises(cia.CtgnError):
            cia.AdaptiveSamplePointsInGrid(gm, raw[:100], cia.AdaptiveGridSamplingOptions(distance_voxel_size=bad))


def test_golden_frame_steps_through_the_gpu(golden_frame_steps):
    """SURVEY 8f rows 1-3 against the committed golden vectors (tests/golden/frame_steps_small.npz, plain-Python restatements of
    the reference): both samplers (same indices, same order), the undistortion loop, and the device-resident map's insert
    decisions / eviction / point sets."""
    g = golden_frame_steps
    raw, t, world = g["raw"], g["t"], g["world"]
    res, min_d, cap = g["map_params"]
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(float(res), float(min_d), int(cap))],
                                                default_radius=0.75, device_updates=True))
    for size in (0.5, 1.5):
        assert np.array_equal(cia.grid_sampling(gm, raw, size), g[f"grid_{size}"])
    assert np.array_equal(cia.AdaptiveSamplePointsInGrid(gm, raw), g["adaptive_default"])
    o = cia.AdaptiveGridSamplingOptions(num_points_per_voxel=2, max_num_points=300)
    assert np.array_equal(cia.AdaptiveSamplePointsInGrid(gm, raw, o), g["adaptive_k2_max300"])
    assert np.abs(cia.transform_points(gm, raw, t, g["pose"], g["tbe"]) - world).max() < 1e-11
    assert np.array_equal(np.asarray(gm.InsertPointCloud(world[:2500]), dtype=bool), g["insert_kept_1"])
    gm.RemoveElementsFarFromLocation(g["remove_loc"], float(g["remove_distance"]))
    assert np.array_equal(_sorted_rows(gm.MapAsPointCloud(0)), _sorted_rows(g["points_after_remove"]))
    assert np.array_equal(np.asarray(gm.InsertPointCloud(world[2500:]), dtype=bool), g["insert_kept_2"])
    assert np.array_equal(_sorted_rows(gm.MapAsPointCloud(0)), _sorted_rows(g["points_final"]))


def test_device_map_fuzz_against_a_dict_model():
    """The fuzz of tests/test_oracle_properties.py::test_map_maintenance_fuzz_against_a_dict_model on the DEVICE-resident map:
    insert decisions, evictions and point sets against the plain-Python dict-of-lists model of map.h:261-293,305-322."""
    from conftest import MAP_FUZZ_LEVELS, load_frame_steps_module, map_fuzz_steps
    mk = load_frame_steps_module()
    for seed, (res, min_d, cap) in enumerate(MAP_FUZZ_LEVELS):
        model = mk.DictMap(res, min_d, cap)
        gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(res, min_d, cap)], default_radius=0.75,
                                                    device_updates=True))
        for step, (pts, evict) in enumerate(map_fuzz_steps(seed, min_d)):
            assert np.array_equal(np.asarray(gm.InsertPointCloud(pts), dtype=bool), model.insert(pts)), (seed, step)
            if evict is not None:
                model.remove_far(evict, 4.0)
                gm.RemoveElementsFarFromLocation(evict, 4.0)
            want_pts = _sorted_rows(model.points())
            assert np.array_equal(_sorted_rows(gm.MapAsPointCloud(0)), want_pts), (seed, step)
            assert gm.NumPoints() == len(want_pts)


def test_sequence_of_frames_end_to_end(street_case):
    """The per-frame loop of Odometry::DoRegister (reference src/ct_icp/odometry.cpp:386-501) with every data-parallel step on
    the GPU — frame grid sampling, keypoint grid sampling, GN registration with the previous-frame motion model, full-scan
    undistortion, map insertion + far-voxel eviction on the device — against the same loop on the CPU oracle, frame by frame
    over a short sequence: identical keypoint sets, insert decisions and map, poses within 1e-7."""
    case = street_case
    res, radius = [(0.8, 0.1, 30)], 0.75
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=radius,
                                                device_updates=True))
    om = orc.Map(resolutions=res, default_radius=radius)
    o = _opts(num_iters_icp=5, threshold_orientation_norm=1e-4, min_number_neighbors=10)
    reg = cia.CT_ICP_Registration(o)
    mm = cia.PreviousFrameMotionModel()
    # frames 0-4 initialise the map with ground-truth poses (the reference's first frames are inserted as they are)
    for j in range(5):
        sc = case["scans"][j]
        keep = cia.grid_sampling(gm, sc.raw, 0.5)
        assert np.array_equal(np.sort(keep), np.sort(orc.grid_sampling(sc.raw, 0.5)))
        keep = np.sort(keep)
        world = cia.transform_points(gm, sc.raw[keep], sc.t[keep], sc.pose_gt, sc.t_begin_end)
        assert np.array_equal(gm.InsertPointCloud(world), om.insert(world))
    prev_g = prev_o = case["scans"][4].pose_gt.copy()
    for j in range(5, 10):
        sc = case["scans"][j]
        keep = np.sort(cia.grid_sampling(gm, sc.raw, 0.5))                               # InitializeFrame (odometry.cpp:349-352)
        raw, t = sc.raw[keep], sc.t[keep]
        kp = np.sort(cia.grid_sampling(gm, raw, 0.7))                                    # TryRegister (odometry.cpp:538)
        assert np.array_equal(kp, np.sort(orc.grid_sampling(raw, 0.7)))
        # constant-velocity prediction from the previous optimised frame, shared by both sides (odometry.cpp:276-330)
        pose0 = syn.perturb_pose(sc.pose_gt, 0.002, 0.02, seed=j)
        world0 = cia.transform_points(gm, raw[kp], t[kp], pose0, sc.t_begin_end)
        mm.previous_frame = cia.TrajectoryFrame.from_pose14(prev_g, 0.0, 0.0)
        kps = np.zeros(len(kp), dtype=cia.WPOINT3D_DTYPE)
        kps["raw_point"], kps["t"], kps["world_point"] = raw[kp], t[kp], world0
        frame = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
        summ = reg.Register(gm, kps, frame, mm)
        op = orc.MotionPrior(previous_begin_tr=prev_o[4:7], previous_end_tr=prev_o[11:14])
        pose_o, _, so = orc.register_gn(om, raw[kp], world0, t[kp], pose0, sc.t_begin_end, _oopts(o), op, heap_mode=0)
        assert summ.success and so.success and summ.num_residuals_used == so.num_residuals_used and summ.num_iters == so.num_iters
        pose_g = frame.pose14()
        tr, rot = se3.pose_error(pose_g, pose_o)
        assert tr < 1e-7 and rot < 1e-7, (j, tr, rot)
        # undistort the whole frame with the optimised pose and update the map (odometry.cpp:461-486, :936-952)
        world_g = cia.transform_points(gm, raw, t, pose_g, sc.t_begin_end)
        world_o = orc.transform_points(pose_o, sc.t_begin_end, t[::50], raw[::50])
        assert np.abs(world_g[::50] - world_o).max() < 1e-6
        # feed BOTH maps the same points so that a 1e-8 pose difference cannot flip an insert decision
        assert np.array_equal(gm.InsertPointCloud(world_g), om.insert(world_g))
        gm.RemoveElementsFarFromLocation(pose_g[11:14], 60.0)
        om.remove_far(pose_g[11:14], 60.0)
        assert gm.NumPoints() == om.num_points() and gm.NumVoxels(0) == om.num_voxels(0)
        prev_g, prev_o = pose_g, pose_o
    assert np.array_equal(_sorted_rows(gm.MapAsPointCloud(0)), _sorted_rows(om.export(0)))


def test_stop_poll_changes_nothing_but_the_launches(street_case):
    """The host watches the solve's stop flag (k_reduce_solve publishes it in mapped host memory) and does not enqueue the iterations behind
    it: a registration that converges early returns the same poses, world points and summary — bit for bit — whether the host polls
    (tuning stop_poll=1, the default) or enqueues all num_iters_icp iterations (0); one that never converges runs its full budget either way."""
    from ct_icp_amd import _lib as L
    case = street_case
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in case["resolutions"]],
                                                default_radius=case["default_radius"]))
    for j in range(7):
        gm.InsertPointCloud(case["scans"][j].world_gt)
    sc = case["scans"][7]
    sel = syn.grid_sample_indices(sc.raw, 0.5)
    pose0 = syn.perturb_pose(sc.pose_gt, 0.003, 0.03, seed=8)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, sc.t[sel], sc.raw[sel])
    results = {}
    try:
        for poll in (1, 0, 1):
            L.lib().ctgn_set_tuning(b"stop_poll", float(poll))
            for thr, iters in ((0.1, 12), (0.0, 6)):
                kps = np.zeros(len(sel), dtype=cia.WPOINT3D_DTYPE)
                kps["raw_point"], kps["t"], kps["world_point"] = sc.raw[sel], sc.t[sel], world0
                frame = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
                summ = cia.CT_ICP_Registration(_opts(num_iters_icp=iters, threshold_orientation_norm=thr)).Register(gm, kps, frame)
                got = (frame.pose14().copy(), kps["world_point"].copy(), summ.num_iters, summ.num_residuals_used, summ.success)
                if (thr, iters) in results:
                    want = results[(thr, iters)]
                    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]) and want[2:] == got[2:], (poll, thr)
                results[(thr, iters)] = got
    finally:
        L.lib().ctgn_set_tuning(b"stop_poll", 1.0)
    assert results[(0.1, 12)][2] < 12 and results[(0.1, 12)][4]          # converged early ...
    assert results[(0.0, 6)][2] == 6                                       # ... and a threshold of zero never stops


def test_device_shuffle_is_a_keyed_permutation_and_equals_that_order_passed_by_the_caller(street_case):
    """ctgn_frame_options::shuffle_seed: the processing order made on the device is a permutation of 0..n-1 (every index once), depends on
    the seed and nothing else, looks like a shuffle (no correlation with the scan order, displacements spread like a uniform permutation's),
    and a frame processed under it is bit for bit the frame processed under the same order passed by the caller as `order`."""
    case = street_case
    res, radius = [(0.8, 0.1, 30)], 0.75
    mk = lambda: cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=radius,
                                                        device_updates=True))
    ga, gb = mk(), mk()
    keep_all = cia.FramePipeline(ga, frame_voxel_size=-1.0, sample_voxel_size=0.7)       # frame_voxel_size <= 0: the sampled frame IS the order
    sc = case["scans"][6]
    n = len(sc.t)
    orders = {}
    for seed in (1, 2, 0x9E3779B97F4A7C15, 1):
        b = keep_all.begin(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, shuffle_seed=seed)
        order = b["sampled_indices"].astype(np.int64)
        assert len(order) == n and np.array_equal(np.sort(order), np.arange(n))
        if seed in orders:
            assert np.array_equal(orders[seed], order)                                 # the seed and n decide it
        orders[seed] = order
        j = np.arange(n)
        assert abs(np.corrcoef(j, order)[0, 1]) < 0.02
        assert 0.30 * n < np.abs(order - j).mean() < 0.37 * n                            # a uniform permutation: n / 3
        assert np.count_nonzero(order == j) < 20
        # neighbours in processing order are not neighbours in the scan: the gaps of consecutive outputs look uniform too
        assert 0.30 * n < np.abs(np.diff(order)).mean() < 0.37 * n
    assert not np.array_equal(orders[1], orders[2]) and np.count_nonzero(orders[1] == orders[2]) < 20
    for m in (1, 2, 3, 5, 64, 257, 4097):                                                # sizes round the Feistel domain's edges
        b = keep_all.begin(sc.raw[:m], sc.t[:m], sc.pose_gt, sc.t_begin_end, shuffle_seed=7)
        assert np.array_equal(np.sort(b["sampled_indices"]), np.arange(m))
    # a registered frame under the device's order == the same frame with that order handed in
    fa = cia.FramePipeline(ga, frame_voxel_size=0.5, sample_voxel_size=0.7)
    fb = cia.FramePipeline(gb, frame_voxel_size=0.5, sample_voxel_size=0.7)
    o0, o = _opts(num_iters_icp=0), _opts(num_iters_icp=5, threshold_orientation_norm=1e-4, min_number_neighbors=10)
    for j in range(5):
        s5 = case["scans"][j]
        for fp in (fa, fb):
            fp.frame(s5.raw, s5.t, s5.pose_gt, s5.t_begin_end, o0, 60.0, want_all=False)
    pose0 = syn.perturb_pose(sc.pose_gt, 0.002, 0.02, seed=6)
    a = fa.register(sc.raw, sc.t, pose0, sc.t_begin_end, o, shuffle_seed=1)
    b = fb.register(sc.raw, sc.t, pose0, sc.t_begin_end, o, order=orders[1].astype(np.uint32))
    assert a["summary"].success and a["summary"].num_iters > 0
    for key in ("sampled_indices", "keypoint_indices", "pose", "sampled_world", "all_world"):
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(fa.update_map(a["pose"][11:14], 60.0, True), fb.update_map(b["pose"][11:14], 60.0, True))
    # ... and it is another choice of surviving points than the scan order's, of the same size (one per occupied voxel)
    c = cia.FramePipeline(mk(), frame_voxel_size=0.5, sample_voxel_size=0.7).begin(sc.raw, sc.t, pose0, sc.t_begin_end)
    assert len(c["sampled_indices"]) == len(a["sampled_indices"]) and not np.array_equal(np.sort(c["sampled_indices"]), np.sort(a["sampled_indices"]))


@pytest.mark.parametrize("shuffled", [False, True])
def test_frame_steps_equal_the_one_call_pipeline(street_case, shuffled):
    """ctgn_frame_begin / ctgn_frame_try_register / ctgn_frame_undistort / ctgn_frame_update_map — the calls integration/odometry_gpu_arm.h makes
    from the reference's InitializeFrame, TryRegister, undistortion loops and UpdateMap — against ctgn_frame_register + ctgn_frame_update_map,
    which the test below pins to the stage calls and those to the oracle: identical sampled frame, keypoints, poses (bit for bit), undistorted
    points, insert masks and maps, GN and robust route; a retry on the same resident frame with another keypoint voxel equals a fresh
    one-call frame at that voxel; the keypoints' world points are the final poses applied to their raw points; undistortion with poses
    the HOST chose (not the registration's) is what the map then receives."""
    case = street_case
    res, radius = [(0.8, 0.1, 30)], 0.75
    mk = lambda: cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=radius,
                                                        device_updates=True))
    ga, gb, gc = mk(), mk(), mk()             # a: one call, b: step by step, c: stage calls
    fa = cia.FramePipeline(ga, frame_voxel_size=0.5, sample_voxel_size=0.7)
    fb = cia.FramePipeline(gb, frame_voxel_size=0.5, sample_voxel_size=0.7)
    o = _opts(num_iters_icp=5, threshold_orientation_norm=1e-4, min_number_neighbors=10)
    o0 = _opts(num_iters_icp=0)
    rng = np.random.default_rng(5)
    for j in range(5):                        # map bootstrap through both spellings (no registration: the undistortion takes pose_gt)
        sc = case["scans"][j]
        order = rng.permutation(len(sc.t)).astype(np.uint32) if shuffled else None
        a = fa.register(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, o0, order=order)
        mask_a = fa.update_map(a["pose"][11:14], 60.0, True)
        b0 = fb.begin(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, order=order, want_world=True)
        assert np.array_equal(a["sampled_indices"], b0["sampled_indices"]) and b0["num_keypoints"] == len(a["keypoint_indices"])
        b2 = fb.undistort(sc.pose_gt, sc.t_begin_end)
        assert np.array_equal(a["sampled_world"], b2["sampled_world"]) and np.array_equal(a["all_world"], b2["all_world"])
        assert np.array_equal(b0["sampled_world"], b2["sampled_world"])          # the initial estimate IS the final pose here
        mask_b = fb.update_map(sc.pose_gt[11:14], 60.0, True)
        assert np.array_equal(mask_a, mask_b)
    prev = case["scans"][4].pose_gt.copy()
    for j in range(5, 10):
        sc = case["scans"][j]
        order = rng.permutation(len(sc.t)).astype(np.uint32) if shuffled else None
        pose0 = syn.perturb_pose(sc.pose_gt, 0.002, 0.02, seed=j)
        mm = cia.PreviousFrameMotionModel()
        mm.previous_frame = cia.TrajectoryFrame.from_pose14(prev, 0.0, 0.0)
        options = o if j != 7 else cia.CTICPOptions(solver=cia.CERES, num_iters_icp=4, ls_max_num_iters=4, min_number_neighbors=10,
                                                    debug_print=False)
        a = fa.register(sc.raw, sc.t, pose0, sc.t_begin_end, options, motion_model=mm, order=order)
        if j % 2:                             # both spellings of the upload: in one call, or ahead of the order (ctgn_frame_stage)
            b0 = fb.begin(sc.raw, sc.t, pose0, sc.t_begin_end, order=order, want_world=(j == 6))
        else:
            fb.stage(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end)
            b0 = fb.begin(None, None, pose0, sc.t_begin_end, order=order, want_world=(j == 6))
        assert np.array_equal(a["sampled_indices"], b0["sampled_indices"])
        if j == 6:                            # InitializeFrame's transform (odometry.cpp:371-375): the sampled frame under the initial estimate
            want = cia.transform_points(ga, sc.raw[b0["sampled_indices"]], sc.t[b0["sampled_indices"]], pose0, sc.t_begin_end)
            assert np.array_equal(b0["sampled_world"], want)
        if j == 8:                            # a first attempt with other settings, then the retry from the initial estimate (odometry.cpp:794-845)
            first = fb.try_register(pose0, sc.t_begin_end, _opts(num_iters_icp=2, min_number_neighbors=10), motion_model=mm, sample_voxel_size=1.1)
            assert first["summary"].success and len(first["keypoint_indices"]) < len(a["keypoint_indices"])
            raw_p = sc.raw if order is None else sc.raw[order]                    # its keypoints = grid_sampling of the sampled frame at that voxel
            keep = cia.grid_sampling(gc, raw_p, 0.5)
            idx = keep if order is None else order[keep]
            assert np.array_equal(first["keypoint_indices"], idx[cia.grid_sampling(gc, raw_p[keep], 1.1)])
        b1 = fb.try_register(pose0, sc.t_begin_end, options, motion_model=mm)
        assert a["summary"].success and b1["summary"].success and b1["summary"].num_iters > 0
        assert a["summary"].num_residuals_used == b1["summary"].num_residuals_used and a["summary"].num_iters == b1["summary"].num_iters
        assert np.array_equal(a["keypoint_indices"], b1["keypoint_indices"]) and np.array_equal(a["pose"], b1["pose"])
        kp = b1["keypoint_indices"]
        assert np.abs(b1["keypoint_world"] - cia.transform_points(ga, sc.raw[kp], sc.t[kp], b1["pose"], sc.t_begin_end)).max() < 1e-9
        with pytest.raises(cia.CtgnError):    # nothing undistorted yet: the map has nothing to take
            fb.update_map(b1["pose"][11:14], 60.0, True)
        b2 = fb.undistort(b1["pose"], sc.t_begin_end)
        assert np.array_equal(a["sampled_world"], b2["sampled_world"]) and np.array_equal(a["all_world"], b2["all_world"])
        if j == 9:                            # the host settles on other poses (e.g. a rejected registration keeps the estimate): those count
            b2 = fb.undistort(pose0, sc.t_begin_end)
            idx = b0["sampled_indices"]
            assert np.array_equal(b2["sampled_world"], cia.transform_points(ga, sc.raw[idx], sc.t[idx], pose0, sc.t_begin_end))
            assert np.array_equal(b2["all_world"], cia.transform_points(ga, sc.raw, sc.t, pose0, sc.t_begin_end))
            fa.register(sc.raw, sc.t, pose0, sc.t_begin_end, o0, order=order, want_all=False)       # the same frame, unregistered, on the one-call side
            a["pose"] = pose0
        mask_a = fa.update_map(a["pose"][11:14], 60.0, True)
        mask_b = fb.update_map(a["pose"][11:14], 60.0, True)
        assert np.array_equal(mask_a, mask_b)
        assert ga.NumPoints() == gb.NumPoints() and ga.NumVoxels(0) == gb.NumVoxels(0)
        prev = b1["pose"]
    assert np.array_equal(_sorted_rows(ga.MapAsPointCloud(0)), _sorted_rows(gb.MapAsPointCloud(0)))
    # error paths: no resident scan on a fresh handle; timestamps outside the poses' interval at undistortion time
    fresh = cia.FramePipeline(mk())
    with pytest.raises(cia.CtgnError):
        fresh.try_register(sc.pose_gt, sc.t_begin_end, o)
    with pytest.raises(cia.CtgnError):
        fresh.undistort(sc.pose_gt, sc.t_begin_end)
    with pytest.raises(cia.CtgnError):
        fb.undistort(sc.pose_gt, (sc.t_begin_end[0] + 0.05, sc.t_begin_end[1]))
    fb._staged_n = len(sc.t)
    with pytest.raises(cia.CtgnError):        # nothing staged ahead
        fb.begin(None, None, sc.pose_gt, sc.t_begin_end)
    fb.stage(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end)
    with pytest.raises(cia.CtgnError):        # a staged scan, but an order that is no permutation (checked on the device)
        fb.begin(None, None, sc.pose_gt, sc.t_begin_end, order=np.zeros(len(sc.t), dtype=np.uint32))


@pytest.mark.parametrize("shuffled", [False, True])
def test_frame_pipeline_equals_the_stage_by_stage_calls(street_case, shuffled):
    """ctgn_frame_register / ctgn_frame_update_map / ctgn_frame (scan resident on the device) against the same
    frame loop spelled with the stage entry points that the test above pins to the oracle: identical sampled frame, keypoints, poses
    (bit for bit: same kernels on the same inputs in the same order), undistorted points, insert decisions and map — with the scan in
    firing order and behind a caller-side shuffle (`order`), for the GN route and once for the robust-loss route."""
    case = street_case
    res, radius = [(0.8, 0.1, 30)], 0.75
    mk = lambda: cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=radius,
                                                        device_updates=True))
    ga, gb = mk(), mk()                       # a: stage by stage, b: frame pipeline
    fp = cia.FramePipeline(gb, frame_voxel_size=0.5, sample_voxel_size=0.7)
    o = _opts(num_iters_icp=5, threshold_orientation_norm=1e-4, min_number_neighbors=10)
    reg = cia.CT_ICP_Registration(o)
    rng = np.random.default_rng(3)

    def stage_frame(sc, pose0, options, mm, order, register=True):
        raw_all, t_all = (sc.raw, sc.t) if order is None else (sc.raw[order], sc.t[order])
        keep = cia.grid_sampling(ga, raw_all, 0.5)
        assert np.all(np.diff(keep.astype(np.int64)) > 0)
        raw, t = raw_all[keep], t_all[keep]
        kp = cia.grid_sampling(ga, raw, 0.7)
        pose = pose0.copy()
        summ = None
        if register:
            kps = np.zeros(len(kp), dtype=cia.WPOINT3D_DTYPE)
            kps["raw_point"], kps["t"] = raw[kp], t[kp]
            kps["world_point"] = cia.transform_points(ga, raw[kp], t[kp], pose0, sc.t_begin_end)
            frame = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
            summ = cia.CT_ICP_Registration(options).Register(ga, kps, frame, mm)
            pose = frame.pose14()
        world = cia.transform_points(ga, raw, t, pose, sc.t_begin_end)
        all_world = cia.transform_points(ga, sc.raw, sc.t, pose, sc.t_begin_end)
        ga.RemoveElementsFarFromLocation(pose[11:14], 60.0)
        mask = ga.InsertPointCloud(world)
        idx = keep if order is None else order[keep]
        return dict(pose=pose, summary=summ, sampled=idx, keypoints=idx[kp], world=world, all_world=all_world, mask=mask)

    def check(a, b, mask_b):
        assert np.array_equal(a["sampled"], b["sampled_indices"]) and np.array_equal(a["keypoints"], b["keypoint_indices"])
        assert np.array_equal(a["pose"], b["pose"])
        assert np.array_equal(a["world"], b["sampled_world"]) and np.array_equal(a["all_world"], b["all_world"])
        if mask_b is not None:
            assert np.array_equal(np.asarray(a["mask"]).astype(bool), mask_b.astype(bool))
        assert ga.NumPoints() == gb.NumPoints() and ga.NumVoxels(0) == gb.NumVoxels(0)

    # frames 0-4: no registration (num_iters_icp = 0 leaves the initial estimate), everything else runs
    o0 = _opts(num_iters_icp=0)
    for j in range(5):
        sc = case["scans"][j]
        order = rng.permutation(len(sc.t)).astype(np.uint32) if shuffled else None
        a = stage_frame(sc, sc.pose_gt, o0, None, order, register=False)
        b = fp.register(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, o0, order=order)
        assert b["summary"].success and b["summary"].num_iters == 0
        mask = fp.update_map(b["pose"][11:14], 60.0, True)
        check(a, b, mask)
    prev = case["scans"][4].pose_gt.copy()
    for j in range(5, 10):
        sc = case["scans"][j]
        order = rng.permutation(len(sc.t)).astype(np.uint32) if shuffled else None
        pose0 = syn.perturb_pose(sc.pose_gt, 0.002, 0.02, seed=j)
        mm = cia.PreviousFrameMotionModel()
        mm.previous_frame = cia.TrajectoryFrame.from_pose14(prev, 0.0, 0.0)
        options = o if j != 7 else cia.CTICPOptions(solver=cia.CERES, num_iters_icp=4, ls_max_num_iters=4, min_number_neighbors=10,
                                                    debug_print=False)
        a = stage_frame(sc, pose0, options, mm, order)
        if j % 2:                                                                 # both spellings of the second half
            b = fp.register(sc.raw, sc.t, pose0, sc.t_begin_end, options, motion_model=mm, order=order)
            mask = fp.update_map(b["pose"][11:14], 60.0, True)
        else:
            b = fp.frame(sc.raw, sc.t, pose0, sc.t_begin_end, options, 60.0, motion_model=mm, order=order, want_sampled=True)
            mask = None
        assert a["summary"].success and b["summary"].success
        assert a["summary"].num_residuals_used == b["summary"].num_residuals_used and a["summary"].num_iters == b["summary"].num_iters
        assert b["summary"].num_iters > 0 and not np.array_equal(b["pose"], pose0)        # and it is a registration, not a no-op
        check(a, b, mask)
        prev = b["pose"]
    assert np.array_equal(_sorted_rows(ga.MapAsPointCloud(0)), _sorted_rows(gb.MapAsPointCloud(0)))
    # error paths: timestamps outside the frame, a bad order, no resident frame on a fresh handle
    sc = case["scans"][9]
    with pytest.raises(cia.CtgnError):
        fp.register(sc.raw, sc.t + 10.0, sc.pose_gt, sc.t_begin_end, o)
    with pytest.raises(cia.CtgnError):
        fp.register(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, o, order=np.full(len(sc.t), len(sc.t), dtype=np.uint32))
    dup = np.arange(len(sc.t), dtype=np.uint32)
    dup[5] = dup[4]                                           # in range, but not a permutation: index 4 twice, index 5 never
    with pytest.raises(cia.CtgnError):
        fp.register(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, o, order=dup)
    with pytest.raises(cia.CtgnError):
        cia.FramePipeline(mk()).update_map(np.zeros(3), 10.0, True)
    # the first frames of a sequence: every point takes the end timestamp (odometry.cpp:357-361)
    b = fp.register(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, o0, override_timestamp=sc.t_begin_end[1], want_sampled=False)
    want = cia.transform_points(ga, sc.raw, np.full(len(sc.t), sc.t_begin_end[1]), sc.pose_gt, sc.t_begin_end)
    assert np.array_equal(b["all_world"], want)


def test_frame_call_with_a_failed_registration_leaves_the_map_to_the_eviction(street_case):
    """Round 4: ctgn_frame enqueues the map update behind the undistortion with the gate read on the device (GnState::failed): a frame
    whose registration fails softly (fewer than 100 keypoints pass, ct_icp.cpp:860-871) must not be inserted — exactly what the two
    separate calls do with add_points = success — while the eviction round the (unchanged) pose still runs."""
    case = street_case
    mk = lambda: cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75, device_updates=True))
    ga, gb = mk(), mk()
    for j in range(4):
        ga.InsertPointCloud(case["scans"][j].world_gt)
        gb.InsertPointCloud(case["scans"][j].world_gt)
    sc = case["scans"][4]
    far = sc.pose_gt.copy(); far[4:7] += 500.0; far[11:14] += 500.0          # nowhere near the map: no neighbours, soft failure
    o = _opts(num_iters_icp=3)
    fa, fb = cia.FramePipeline(ga, 0.5, 1.5), cia.FramePipeline(gb, 0.5, 1.5)
    ra = fa.register(sc.raw, sc.t, far, sc.t_begin_end, o, want_all=False, want_sampled=False)
    assert not ra["summary"].success
    fa.update_map(ra["pose"][11:14], 60.0, False)
    rb = fb.frame(sc.raw, sc.t, far, sc.t_begin_end, o, 60.0, want_all=False, want_sampled=False)
    assert not rb["summary"].success and np.array_equal(ra["pose"], rb["pose"])
    assert ga.NumPoints() == gb.NumPoints() == 0              # everything lies farther than 60 m from the failed frame's pose: evicted, nothing inserted
    # and a frame that succeeds on the same handle afterwards is inserted
    for g in (ga, gb):
        for j in range(4):
            g.InsertPointCloud(case["scans"][j].world_gt)
    pose0 = syn.perturb_pose(sc.pose_gt, 0.002, 0.02, seed=5)
    ra = fa.register(sc.raw, sc.t, pose0, sc.t_begin_end, o, want_all=False, want_sampled=False)
    fa.update_map(ra["pose"][11:14], 60.0, True)
    rb = fb.frame(sc.raw, sc.t, pose0, sc.t_begin_end, o, 60.0, want_all=False, want_sampled=False)
    assert ra["summary"].success and rb["summary"].success and np.array_equal(ra["pose"], rb["pose"])
    assert ga.NumPoints() == gb.NumPoints() > 0
    pa, pb = ga.MapAsPointCloud(), gb.MapAsPointCloud()
    assert np.array_equal(pa[np.lexsort(pa.T[::-1])], pb[np.lexsort(pb.T[::-1])])


def test_page_locked_scan_arrays_are_read_and_written_in_place(street_case):
    """Round 4: a frame call whose scan rows, timestamps and output array are page-locked host memory (ct_icp_amd.pinned_array) skips
    the staging copy on the way in and the hand-over copy on the way out (ctgn_frame_register: DMA from / to the caller's arrays, the
    x y z t records written by a kernel). Same poses, indices, world points and map as the staged path, bit for bit — also with the
    first frames' overridden timestamp, with only some of the arrays page-locked, and with a timestamp outside the frame's interval."""
    case = street_case
    mk = lambda: cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(0.8, 0.1, 30)], default_radius=0.75, device_updates=True))
    ga, gb = mk(), mk()
    for j in range(4):
        ga.InsertPointCloud(case["scans"][j].world_gt)
        gb.InsertPointCloud(case["scans"][j].world_gt)
    sc = case["scans"][4]
    n = len(sc.t)
    raw_p, t_p, out_p = cia.pinned_array((n, 3)), cia.pinned_array(n), cia.pinned_array((n, 3))
    raw_p[:] = sc.raw
    t_p[:] = sc.t
    out_p[:] = -1.0
    pose0 = syn.perturb_pose(sc.pose_gt, 0.002, 0.02, seed=5)
    o = _opts(num_iters_icp=4)
    fa, fb = cia.FramePipeline(ga, 0.5, 1.5), cia.FramePipeline(gb, 0.5, 1.5)
    ra = fa.register(sc.raw, sc.t, pose0, sc.t_begin_end, o, want_all=True, want_sampled=True)
    rb = fb.register(raw_p, t_p, pose0, sc.t_begin_end, o, want_all=True, want_sampled=True, all_world_out=out_p)
    assert rb["all_world"] is out_p
    for key in ("pose", "sampled_indices", "keypoint_indices", "all_world", "sampled_world"):
        assert np.array_equal(ra[key], rb[key]), key
    # mixed: page-locked input, pageable output and the other way round
    rc = fb.register(raw_p, t_p, pose0, sc.t_begin_end, o, want_all=True, want_sampled=False)
    out_p[:] = -1.0
    rd = fb.register(sc.raw, sc.t, pose0, sc.t_begin_end, o, want_all=True, want_sampled=False, all_world_out=out_p)
    assert np.array_equal(rc["all_world"], ra["all_world"]) and np.array_equal(rd["all_world"], ra["all_world"])
    # one timestamp for every point (the first two registered frames, odometry.cpp:357-361)
    o0 = _opts(num_iters_icp=0)
    ea = fa.register(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, o0, override_timestamp=sc.t_begin_end[1], want_sampled=False)
    eb = fb.register(raw_p, t_p, sc.pose_gt, sc.t_begin_end, o0, override_timestamp=sc.t_begin_end[1], want_sampled=False, all_world_out=out_p)
    assert np.array_equal(ea["all_world"], eb["all_world"])
    # the whole frame, map update included
    fa_ = fa.frame(sc.raw, sc.t, pose0, sc.t_begin_end, o, 60.0, want_all=True, want_sampled=False)
    fb_ = fb.frame(raw_p, t_p, pose0, sc.t_begin_end, o, 60.0, want_all=True, want_sampled=False, all_world_out=out_p)
    assert np.array_equal(fa_["pose"], fb_["pose"]) and np.array_equal(fa_["all_world"], fb_["all_world"])
    assert ga.NumPoints() == gb.NumPoints() > 0
    pa, pb = ga.MapAsPointCloud(), gb.MapAsPointCloud()
    assert np.array_equal(pa[np.lexsort(pa.T[::-1])], pb[np.lexsort(pb.T[::-1])])
    # a timestamp outside [t_begin, t_end] (or NaN) is refused as on the staged path, and the handle goes on working
    for bad in (sc.t_begin_end[1] + 1.0, np.nan):
        t_p[n // 2] = bad
        with pytest.raises(cia.CtgnError):
            fb.register(raw_p, t_p, pose0, sc.t_begin_end, o, want_all=False, want_sampled=False)
    t_p[:] = sc.t
    rz = fb.register(raw_p, t_p, pose0, sc.t_begin_end, o, want_all=False, want_sampled=False)
    assert rz["summary"].success


def test_map_or_keypoints_changed_inside_a_stepwise_solve_drop_the_carried_bound(street_case):
    """Every search after the first of a solve is bounded by the previous search's k-th neighbour distance —
    valid only while map and keypoints stay what they were. The stepwise API lets a caller change either between two accumulate
    calls: evicting the voxels around the keypoints, or uploading other keypoints, must make the next search start from the radius
    again. Checked on the per-keypoint neighbour counts and farthest neighbours against the batched radius search at the world points
    the search used, on the map as it then is."""
    case = street_case
    om, gm = build_maps(case, 6, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(case, 6, 0.6)
    o = _opts(num_iters_icp=8, threshold_orientation_norm=0.0, min_number_neighbors=5)
    s = cia.GnSolver(gm)
    s.set_debug(True)

    def check(tag):
        dbg = s.get_debug()
        nbs = gm.ComputeNeighborhoods(s.world_points(), 20)
        want_n = np.array([len(g) for g in nbs])
        assert np.array_equal(dbg["n_neighbors"], want_n), (tag, int(np.count_nonzero(dbg["n_neighbors"] != want_n)))
        for i in np.flatnonzero(want_n > 0)[::7]:
            assert np.array_equal(dbg["farthest"][i], nbs[i][0]), (tag, i)                 # farthest first (map.h:508-513)
        return want_n

    s.set_keypoints(raw, world0, t)
    s.gn_begin(pose0, sc.t_begin_end, o)
    for _ in range(2):
        s.gn_accumulate()
        s.gn_solve_update()
    full = check("bounded search, nothing changed")
    # thin the map out under the keypoints: every voxel farther than 6 m from a point beside the sensor goes
    gm.RemoveElementsFarFromLocation(sc.pose_gt[11:14] + np.array([3.0, 0.0, 0.0]), 6.0)
    s.gn_accumulate()
    thinned = check("after an eviction inside the solve")
    assert (thinned < full).sum() > 100                                           # the eviction did bite
    s.gn_solve_update()
    # other keypoints inside the running solve
    s.set_keypoints(raw[1::2], world0[1::2], t[1::2])
    s.gn_accumulate()
    check("after new keypoints inside the solve")
    s.gn_solve_update()
    s.gn_end()


def test_frame_pipeline_edge_cases(street_case):
    """ctgn_frame_register on the inputs a caller can throw at it: an empty scan, no sub-sampling at all (every point a keypoint),
    the keypoint cap, float32 and strided (WPoint3D-shaped) views through the C ABI directly, and a second call on the same handle with
    a larger scan (scratch growth)."""
    import ctypes as C
    from ct_icp_amd import _lib as L
    case = street_case
    res, radius = [(0.8, 0.1, 30)], 0.75
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=radius,
                                                device_updates=True))
    o = _opts(num_iters_icp=3, min_number_neighbors=10)
    o0 = _opts(num_iters_icp=0)
    fp = cia.FramePipeline(gm, frame_voxel_size=0.5, sample_voxel_size=0.7)
    for j in range(4):                                                     # a map to register against
        sc = case["scans"][j]
        fp.frame(sc.raw, sc.t, sc.pose_gt, sc.t_begin_end, o0, 60.0, want_all=False)
    sc = case["scans"][4]
    # empty scan: a soft failure of the registration (no keypoints), nothing inserted, the map untouched
    before = gm.NumPoints()
    r = fp.register(np.zeros((0, 3)), np.zeros(0), sc.pose_gt, sc.t_begin_end, o)
    assert not r["summary"].success and len(r["sampled_indices"]) == 0 and r["all_world"].shape == (0, 3)
    fp.update_map(sc.pose_gt[11:14], 1e9, True)                        # nothing to evict at this distance, nothing to insert
    assert gm.NumPoints() == before
    # a small scan first, then the full one on the same handle (the frame scratch grows), no sub-sampling: every point is a keypoint
    fp_all = cia.FramePipeline(gm, frame_voxel_size=0.0, sample_voxel_size=0.0)
    sub = np.sort(cia.grid_sampling(gm, sc.raw, 1.0))
    pose0 = syn.perturb_pose(sc.pose_gt, 0.002, 0.02, seed=3)
    r = fp_all.register(sc.raw[sub], sc.t[sub], pose0, sc.t_begin_end, o)
    assert np.array_equal(r["sampled_indices"], np.arange(len(sub))) and np.array_equal(r["keypoint_indices"], np.arange(len(sub)))
    s = cia.GnSolver(gm)
    s.set_keypoints(sc.raw[sub], cia.transform_points(gm, sc.raw[sub], sc.t[sub], pose0, sc.t_begin_end), sc.t[sub])
    pose_ref, summ_ref, _ = s.solve(pose0, sc.t_begin_end, o)
    assert np.array_equal(r["pose"], pose_ref) and r["summary"].num_residuals_used == summ_ref.num_residuals_used
    # the keypoint cap keeps the first max_num_keypoints keypoints (odometry.cpp:549-552)
    fp_cap = cia.FramePipeline(gm, frame_voxel_size=0.5, sample_voxel_size=0.7, max_num_keypoints=300)
    full = fp.register(sc.raw, sc.t, pose0, sc.t_begin_end, o)
    capped = fp_cap.register(sc.raw, sc.t, pose0, sc.t_begin_end, o)
    assert len(full["keypoint_indices"]) > 300 and np.array_equal(capped["keypoint_indices"], full["keypoint_indices"][:300])
    assert np.array_equal(capped["sampled_indices"], full["sampled_indices"])
    # float32 xyz + float32 t inside 64-byte records (a WPoint3D-like layout), straight through the C ABI
    n = len(sc.t)
    rec = np.zeros(n, dtype=np.dtype({"names": ["xyz", "t", "pad"], "formats": [("<f4", 3), "<f4", ("<f4", 12)], "itemsize": 64}))
    rec["xyz"], rec["t"] = sc.raw.astype(np.float32), sc.t.astype(np.float32)
    t32 = rec["t"].astype(np.float64)
    tbe = np.array([min(sc.t_begin_end[0], t32.min()), max(sc.t_begin_end[1], t32.max())])
    fo = L.FrameOptions(0.5, 0.7, -1, 0, 0.0, 0)
    out = L.FrameOutputs()
    allw = np.zeros((n, 3), dtype=np.float32)
    out.all_world_base, out.all_world_stride_bytes, out.all_world_dtype = allw.ctypes.data, 12, L.CTGN_F32
    pose = pose0.copy()
    summ = L.Summary()
    dp = C.POINTER(C.c_double)
    c_o = cia.registration._c_options(o)
    st = L.lib().ctgn_frame_register(gm.handle, L.View(rec.ctypes.data, 64, L.CTGN_F32, 0), L.View(rec.ctypes.data + 12, 64, L.CTGN_F32, 0), n, None,
                                     C.byref(fo), pose.ctypes.data_as(dp), tbe.ctypes.data_as(dp), C.byref(c_o), None, None, None,
                                     C.byref(out), C.byref(summ))
    L.check(gm.handle, st)
    want = fp.register(rec["xyz"].astype(np.float64), t32, pose0, tbe, o)
    assert np.array_equal(pose, want["pose"]) and out.num_keypoints == len(want["keypoint_indices"])
    assert np.array_equal(allw, want["all_world"].astype(np.float32))


def test_config_e_two_sequences_on_one_gpu(street_case):
    """SURVEY.md 8d config E (one sequence per GPU, zero communication) through ct_icp_amd.sequence_runner on ONE GPU with two
    sequences: every frame is one ctgn_frame call; each sequence owns its map and handle, so a sequence's trajectory does not depend
    on what else ran in the process (bit-identical alone and in the batch); every registration succeeds and tracks the ground truth."""
    from ct_icp_amd import sequence_runner as sr
    case = street_case
    rng = np.random.default_rng(11)
    seq_a = [(sc.raw, sc.t, tuple(sc.t_begin_end)) for sc in case["scans"][:11]]
    seq_b = [(sc.raw, sc.t, tuple(sc.t_begin_end)) for sc in case["scans"][:8]]
    orders_b = [rng.permutation(len(sc.t)).astype(np.uint32) for sc in case["scans"][:8]]
    gt = [sc.pose_gt for sc in case["scans"]]
    kw = dict(device=0, solver=cia.GN, voxel_size=0.5, sample_voxel_size=0.7, max_distance=60.0, init_poses=gt, init_frames=5)
    alone = sr.run_sequence(seq_a, **kw)
    assert alone["frames"] == 11 and alone["success"].all() and alone["map_points"] > 10_000
    out = sr.run_batch({0: seq_a, 1: seq_b}, [len(seq_a), len(seq_b)], rank=0, world_size=1, **kw)
    assert out["frames"] == 19 and out["shares"] == [[0, 1]] and out["frames_per_sec"] > 0
    ra, rb = out["results"]
    assert ra["sequence"] == 0 and np.array_equal(ra["poses"], alone["poses"]) and ra["map_points"] == alone["map_points"]
    assert rb["frames"] == 8 and rb["success"].all()
    for j in range(5, 11):
        tr, rot = se3.pose_error(ra["poses"][j], gt[j])
        assert tr < 0.6 and rot < 5e-3, (j, tr, rot)          # a street canyon: the along-street translation is weakly observed
    # the shuffled copy of the same scans tracks the same trajectory (another random choice of voxel representatives)
    rb2 = sr.run_sequence(seq_b, orders=orders_b, **kw)
    for j in range(5, 8):
        tr, rot = se3.pose_error(rb2["poses"][j], rb["poses"][j])
        assert tr < 0.3 and rot < 5e-3, (j, tr, rot)


def test_config_e_two_gloo_ranks_share_one_gpu():
    """Config E as SURVEY.md 8d defines it (11 sequences, seeds 10-20, lengths proportional to the KITTI sequences, dealt longest first) at
    1/400 of the lengths with a coarse scan pattern, as TWO gloo ranks on the one GPU: each rank generates and runs only its share, one
    ctgn_frame call per frame, nothing is exchanged while they run; the gathered result holds every frame once and no registration fails.
    (The size the survey defines runs through scripts/sequence_run.py --config-e: profiles/r04_config_e_n1.json.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29567",
           os.path.join(root, "tests", "config_e_worker.py")]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["sequences"] == list(range(11)) and d["frames"] == sum(d["lengths"]) and len(d["shares"]) == 2
    assert d["failures"] == 0 and d["err_max"] < 1.0, d


# ------------------------------------------------------------------------------------------------- device-memory views
def test_device_memory_views_match_host_views(street_case):
    """Every entry point that takes point views also takes them in device memory (torch CUDA tensors here): identical results,
    nothing staged through the host."""
    import torch
    case = street_case
    dev = torch.device("cuda", 0)
    res, radius = [(0.8, 0.1, 30)], 0.75
    opts = dict(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=radius, device_updates=True)
    gh, gd = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(**opts)), cia.GpuVoxelMap(cia.GpuVoxelMapOptions(**opts))
    for j in range(4):
        sc = case["scans"][j]
        keep_h = np.sort(cia.grid_sampling(gh, sc.raw, 0.5))
        raw_d, t_d = torch.from_numpy(sc.raw).to(dev), torch.from_numpy(sc.t).to(dev)
        keep_d = torch.sort(cia.grid_sampling(gd, raw_d, 0.5).long()).values
        assert np.array_equal(keep_h, keep_d.cpu().numpy())
        world_h = cia.transform_points(gh, sc.raw[keep_h], sc.t[keep_h], sc.pose_gt, sc.t_begin_end)
        world_d = cia.transform_points(gd, raw_d[keep_d], t_d[keep_d], sc.pose_gt, sc.t_begin_end)
        assert world_d.is_cuda and np.array_equal(world_h, world_d.cpu().numpy())
        kept_h = gh.InsertPointCloud(world_h)
        kept_d = gd.InsertPointCloud(world_d)
        assert np.array_equal(kept_h, kept_d.cpu().numpy())
    assert gh.NumPoints() == gd.NumPoints() and np.array_equal(_sorted_rows(gh.MapAsPointCloud(0)), _sorted_rows(gd.MapAsPointCloud(0)))
    # registration with resident device keypoints; float32 strided device view (a (N, 8) tensor, xyz in columns 0..2, t in 3)
    sc, raw, t, pose0, world0 = _keypoints(case, 4, 0.6)
    o = _opts(num_iters_icp=4, min_number_neighbors=10)
    sh, sd = cia.GnSolver(gh), cia.GnSolver(gd)
    sh.set_keypoints(raw, world0, t)
    sd.set_keypoints(torch.from_numpy(raw).to(dev), torch.from_numpy(world0).to(dev), torch.from_numpy(t).to(dev))
    ph, smh, _ = sh.solve(pose0, sc.t_begin_end, o)
    pd, smd, _ = sd.solve(pose0, sc.t_begin_end, o)
    assert np.array_equal(ph, pd) and smh.num_residuals_used == smd.num_residuals_used
    out_d = torch.empty((len(t), 3), dtype=torch.float64, device=dev)
    assert np.array_equal(sd.world_points(out_d).cpu().numpy(), sh.world_points())
    packed = torch.zeros((len(t), 8), dtype=torch.float32, device=dev)
    packed[:, 0:3] = torch.from_numpy(raw.astype(np.float32)).to(dev)
    w32 = cia.transform_points(gd, packed[:, 0:3], torch.from_numpy(t).to(dev), pose0, sc.t_begin_end)
    w32_h = cia.transform_points(gh, raw.astype(np.float32).astype(np.float64), t, pose0, sc.t_begin_end)
    assert np.array_equal(w32.cpu().numpy(), w32_h)
    # mixing host and device views in one call is refused; a timestamp outside the frame is still caught on the device
    with pytest.raises(cia.CtgnError) as e:
        sd.set_keypoints(torch.from_numpy(raw).to(dev), torch.from_numpy(world0).to(dev), torch.from_numpy(t))
    assert e.value.status == L.ERR_UNSUPPORTED
    t_bad = torch.from_numpy(t).to(dev).clone()
    t_bad[5] = sc.t_begin_end[1] + 1.0
    with pytest.raises(cia.CtgnError) as e:
        cia.transform_points(gd, torch.from_numpy(raw).to(dev), t_bad, pose0, sc.t_begin_end)
    assert e.value.status == L.ERR_TIMESTAMP_RANGE


# ------------------------------------------------------------------------------------------------- config A
def test_config_a_reference_scene_gpu_vs_oracle(config_a_case):
    """BASELINE.json configs[0] (the reference's synthetic courtyard, 0.5 m map, 125 voxels per query, 30 GN iterations): the GPU
    registration equals the oracle's on both solver routes."""
    case = config_a_case
    om, gm = build_maps(case, 4, with_gpu=True)
    sc = case["scans"][4]
    sel = np.sort(syn.grid_sample_indices(sc.raw, case["sample_voxel_size"]))
    raw, t = sc.raw[sel], sc.t[sel]
    pose0 = syn.perturb_pose(sc.pose_gt, 0.01, 0.05, seed=1)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, t, raw)
    o = _opts(num_iters_icp=30, threshold_orientation_norm=1e-6)
    kps = np.zeros(len(t), dtype=cia.WPOINT3D_DTYPE)
    kps["raw_point"], kps["t"], kps["world_point"] = raw, t, world0
    frame = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
    summ = cia.CT_ICP_Registration(o).Register(gm, kps, frame)
    pose_o, world_o, so = orc.register_gn(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), None, heap_mode=0)
    tr, rot = se3.pose_error(frame.pose14(), pose_o)
    assert summ.success and so.success and summ.num_iters == so.num_iters and summ.num_residuals_used == so.num_residuals_used
    assert tr < 1e-7 and rot < 1e-7, (tr, rot)
    assert np.abs(kps["world_point"] - world_o).max() < 1e-6
    # Robust route. The courtyard has poles: neighbourhoods of exactly collinear points, whose two small singular values are both ~0,
    # so the "normal" is whatever vector the roundings of the SVD leave in V[:, 2]. GN gives such blocks the weight a2D^2 = 0, the
    # CERES route weight_neighborhood * exp(..) ~ 0.09: they count. Product, oracle and oracle/_ref therefore all run the SAME
    # statement of Eigen's JacobiSVD in the same (unfused, correctly rounded) arithmetic on the same sums -> the normals agree
    # bit for bit on EVERY block, degenerate ones included, and the end-to-end poses to the stated tolerance and far below.
    s = cia.GnSolver(gm)
    ro = cia.CTICPOptions(solver=cia.CERES, debug_print=False, num_iters_icp=1, ls_max_num_iters=5)
    s.set_keypoints(raw, np.zeros_like(raw), t)
    pose_g, summ_r, _ = s.solve_robust(pose0, sc.t_begin_end, ro)
    got = s.robust_blocks()
    q0 = pose0.copy()
    q0[0:4] /= np.linalg.norm(q0[0:4]); q0[7:11] /= np.linalg.norm(q0[7:11])
    oro = orc.RobustOptions(num_iters_icp=1, ls_max_num_iters=5)
    want = orc.robust_build(om, raw, orc.transform_points(q0, sc.t_begin_end, t, raw), t, sc.t_begin_end, oro, heap_mode=0)
    kp = want["keypoint"]
    assert summ_r.num_residuals_used == len(kp) and np.array_equal(np.nonzero(got["rank"] >= 0)[0], kp)
    assert np.array_equal(got["ref"][kp], want["ref"])
    degenerate = want["weight"] < 0.15                               # a2D ~ 0: only the neighbourhood term is left
    assert degenerate.sum() > 20 and (~degenerate).sum() > 1200
    assert np.array_equal(got["normal"][kp], want["normal"])          # every block, collinear poles included
    assert np.abs(got["weight"][kp] - want["weight"]).max() < 1e-12
    mine = dict(raw=raw[kp], ref=got["ref"][kp], normal=got["normal"][kp], weight=got["weight"][kp], alpha=got["alpha"][kp])
    pose_fixed, _ = orc.robust_solve_fixed(mine, oro, None, q0, 5)
    tr, rot = se3.pose_error(pose_g, pose_fixed)
    assert tr < 1e-8 and rot < 1e-8, (tr, rot)
    ro.num_iters_icp, oro.num_iters_icp = 30, 30
    frame_r = cia.TrajectoryFrame.from_pose14(pose0, *sc.t_begin_end)
    summ_r = cia.CT_ICP_Registration(ro).Register(gm, kps, frame_r)
    pose_ro, _, sro = orc.register_robust(om, raw, t, pose0, sc.t_begin_end, oro, None, heap_mode=0)
    tr, rot = se3.pose_error(frame_r.pose14(), pose_ro)
    assert summ_r.success and sro.success and summ_r.num_iters == sro.num_iters
    assert tr < POSE_TOL_M and rot < POSE_TOL_RAD, (tr, rot)           # the stated tolerance ...
    assert tr < 1e-6 and rot < 1e-6, (tr, rot)                          # ... and what identical normals actually deliver


# ------------------------------------------------------------------------------------------------- BASELINE sizes vs the oracle
@pytest.mark.parametrize("search_kernel", ["rows", "rows_ordered"])
def test_full_size_b2_sweep_matches_oracle(config_b_full, search_kernel):
    """BASELINE.json configs[1] at its full size (the workload bench.py times): every return of the 64-beam sweep (~132 k
    keypoints) against the oracle running OpenMP over keypoints — one accumulation: neighbour counts, farthest neighbours and
    gate decisions identical, packed system <= 1e-10 relative; five iterations: pose <= 1e-7 (stated bar 1e-4). For the row
    kernel in caller order and in home-voxel order."""
    import os
    gm, sc = config_b_full
    om = _B2_ORACLE_MAP["om"]
    n = len(sc.t)
    assert n > 100_000
    threads = max(1, min(16, os.cpu_count() or 1))
    pose0 = syn.perturb_pose(sc.pose_gt, 0.003, 0.03, seed=4)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, sc.t, sc.raw)
    o1 = _opts(num_iters_icp=1, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    s.set_ordering(0 if search_kernel == "rows" else 1)
    s.set_debug(True)
    s.set_keypoints(sc.raw, world0, sc.t)
    pose1, summ, _ = s.solve(pose0, sc.t_begin_end, o1)
    dbg = s.get_debug()
    A, b, n_used = s.get_system()
    Ao, bo, no, info = orc.gn_accumulate(om, sc.raw, world0, sc.t, pose0, sc.t_begin_end, _oopts(o1), heap_mode=0, num_threads=threads, debug=True)
    assert np.array_equal(dbg["n_neighbors"], info["n_neighbors"])
    has = info["n_neighbors"] >= 20
    assert has.sum() > 50_000
    assert np.array_equal(dbg["farthest"][has], info["farthest"][has])
    assert np.array_equal(dbg["used"], info["used"]) and n_used == no == summ.num_residuals_used > 20_000
    assert np.abs(dbg["a2d"][has] - info["a2d"][has]).max() < 1e-9
    scale = np.abs(Ao).max()
    assert np.abs(A - Ao).max() < 1e-10 * scale and np.abs(b - bo).max() < 1e-10 * np.abs(bo).max() + 1e-14
    # five iterations without debug capture (the path bench.py times), with the motion prior
    k = syn.driving_trajectory(12, seed=0, start_x=20.0)
    mm = cia.PreviousFrameMotionModel()
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(np.concatenate([k[9], k[10]]), 0.0, 0.0)
    op = orc.MotionPrior(previous_begin_tr=k[9, 4:7], previous_end_tr=k[10, 4:7])
    o5 = _opts(num_iters_icp=5, threshold_orientation_norm=0.0)
    s.set_debug(False)
    s.set_keypoints(sc.raw, world0, sc.t)
    pose5, summ5, _ = s.solve(pose0, sc.t_begin_end, o5, mm)
    pose_o, world_o, so = orc.register_gn(om, sc.raw, world0, sc.t, pose0, sc.t_begin_end, _oopts(o5), op, heap_mode=0, num_threads=threads)
    assert summ5.success and so.success and summ5.num_iters == so.num_iters == 5
    assert summ5.num_residuals_used == so.num_residuals_used
    tr, rot = se3.pose_error(pose5, pose_o)
    assert tr < 1e-7 and rot < 1e-7, (tr, rot)
    assert np.abs(s.world_points() - world_o).max() < 1e-7
    # at this size the later iterations run on neighbour pools: searching every iteration instead gives the same bits
    w5 = s.world_points()
    s.set_pools(0)
    s.set_keypoints(sc.raw, world0, sc.t)
    pose5_np, summ5_np, _ = s.solve(pose0, sc.t_begin_end, o5, mm)
    assert np.array_equal(pose5, pose5_np) and np.array_equal(w5, s.world_points()) and summ5.num_residuals_used == summ5_np.num_residuals_used


@pytest.mark.parametrize("mode", ["rows", "rows_plain_rank", "lane"])
def test_exact_distance_ties_on_a_lattice_map(mode):
    """map.h:491-500 keeps candidates in a std::priority_queue keyed by the distance only: which of two EQUAL distances survives, and in
    which order equal ones are drained, is decided by libstdc++'s heap layout. Round 3: the product reproduces that. The lane kernel and
    the batched RadiusSearch ARE the reference's queue (heap restated move for move); the row kernel detects (near-)tied candidates in its
    exact rank and replays the reference's queue for that keypoint. On a lattice — dozens of candidates at exactly equal distances around
    every query — neighbour lists, farthest neighbours, gate decisions and normals must equal the oracle's heap_mode 0 (= the reference:
    tests/test_oracle_vs_ref.py pins that bit for bit) and, where oracle/_ref is built, the reference's own RadiusSearch."""
    g = np.arange(-8, 9) * 0.25
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    lattice = lattice[np.random.default_rng(5).permutation(len(lattice))]
    res = [(0.5, 0.01, 40)]
    om = orc.Map(resolutions=res, default_radius=0.8)
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=0.8))
    assert np.array_equal(om.insert(lattice), gm.InsertPointCloud(lattice))
    rng = np.random.default_rng(6)
    core = lattice[np.abs(lattice).max(axis=1) <= 1.0]
    qs = np.concatenate([core[:150], core[:150] + 0.125, core[150:300] + [0.125, 0.0, 0.0], core[:100] + rng.normal(0, 1e-3, (100, 3))])
    if mode == "rows":
        # batched RadiusSearch (ISlamMap::RadiusSearch / ComputeNeighborhoods): the reference's queue
        from oracle import ref as oref
        rmap = None
        if oref.available():
            rmap = oref.Map(resolutions=res, default_radius=0.8)
            rmap.insert(lattice)
        differs_from_total_order = 0
        for k in (20, 8):
            got = gm.ComputeNeighborhoods(qs, k)
            rc, rx = rmap.radius_search(qs, 0.0, k) if rmap is not None else (None, None)
            for j, (q, gq) in enumerate(zip(qs, got)):
                want = om.radius_search(q, 0.0, k, heap_mode=0)
                assert np.array_equal(gq, want)
                if rmap is not None:
                    assert rc[j] == len(gq) and np.array_equal(rx[j, :rc[j]], gq)
                differs_from_total_order += not np.array_equal(want, om.radius_search(q, 0.0, k, heap_mode=1))
        assert differs_from_total_order > 100               # the lattice does separate the two orders
    # the GN kernels on the same queries as keypoints (identity pose, raw = world)
    n = len(qs)
    pose = np.array([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0], float)
    tt = np.linspace(0.0, 1.0, n)
    o = _opts(num_iters_icp=1, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    s.set_variant({"rows": 0, "rows_plain_rank": 2, "lane": 1}[mode])
    s.set_debug(True)
    s.set_keypoints(qs, qs, tt)
    s.solve(pose, (0.0, 1.0), o)
    dbg = s.get_debug()
    Ao, bo, no, info = orc.gn_accumulate(om, qs, qs, tt, pose, (0.0, 1.0), _oopts(o), heap_mode=0, debug=True)
    assert np.array_equal(dbg["n_neighbors"], info["n_neighbors"]) and (info["n_neighbors"] == 20).all()
    assert np.array_equal(dbg["farthest"], info["farthest"])
    assert np.array_equal(dbg["used"], info["used"])
    # (no bit-comparison of the GN normals here: every lattice neighbourhood is symmetric, two singular values tie exactly and the normal is
    # then whatever the eigen-solver's roundings make of it — a2D = 0 gives such a keypoint weight zero; the robust route below, which
    # runs Eigen's JacobiSVD restated operation for operation, IS compared bit for bit)
    A, b, n_used = s.get_system()
    assert n_used == no and np.abs(A - Ao).max() <= 1e-10 * max(np.abs(Ao).max(), 1e-300)
    if mode in ("rows", "rows_plain_rank"):
        # the robust route reads the same neighbour records (k_robust_prepare): its blocks carry the first THREE neighbours of the
        # reference's order (num_closest_neighbors 3) and the normal of the exactly restated JacobiSVD — on a lattice the sums are exact,
        # so the normals are bit-identical iff the kept SETS are
        ro = cia.CTICPOptions(solver=cia.CERES, debug_print=False, min_number_neighbors=20, num_iters_icp=1, ls_max_num_iters=0, num_closest_neighbors=3)
        oro = orc.RobustOptions(ro.num_iters_icp, ro.min_number_neighbors, ro.max_number_neighbors, False, ro.max_num_residuals, ro.loss_function,
                                ro.ls_max_num_iters, ro.num_closest_neighbors, ro.weight_alpha, ro.weight_neighborhood, ro.power_planarity,
                                ro.max_dist_to_plane_ct_icp, ro.ls_sigma, ro.ls_tolerant_min_threshold, ro.threshold_orientation_norm,
                                ro.threshold_translation_norm)
        s.set_debug(False)
        s.set_keypoints(qs, np.zeros_like(qs), tt)
        s.solve_robust(pose, (0.0, 1.0), ro)
        want = orc.robust_build(om, qs, qs, tt, (0.0, 1.0), oro, heap_mode=0)
        got = s.robust_blocks()
        kpi = want["keypoint"][::3]
        assert len(kpi) == n and np.array_equal(kpi, np.arange(n))
        assert np.array_equal(got["normal"], want["normal"][::3])
        gref = got["ref"]                                    # (n, 3): the first reference point; robust_blocks returns block 0 of each keypoint
        assert np.array_equal(gref, want["ref"][::3])
        s.set_debug(True)
    if mode == "rows":
        # a second iteration searches with a carried-over bound: the replay still sees the whole radius
        o2 = _opts(num_iters_icp=3, threshold_orientation_norm=0.0)
        s.set_keypoints(qs, qs, tt)
        pose_g, summ, _ = s.solve(pose, (0.0, 1.0), o2)
        pose_o, _, so = orc.register_gn(om, qs, qs, tt, pose, (0.0, 1.0), _oopts(o2), None, heap_mode=0)
        assert summ.num_residuals_used == so.num_residuals_used and np.abs(pose_g - pose_o).max() < 1e-9


@pytest.mark.parametrize("mode", ["rows", "rows_ordered"])
def test_km_scale_world_coordinates(street_case, mode):
    """SURVEY section 7 'Precision': the same scene 5 km from the origin (absolute FP64 coordinates, as the reference stores them).
    Discrete results stay identical to the oracle; the covariance C = SS / n - mu mu^T loses ~|p|^2 / sigma^2 * eps of relative
    accuracy in ANY summation order (the reference's included), so a2D / normals / system are compared at the bar that conditioning
    allows, the pose at 1e-6 (stated tolerance 1e-4)."""
    shift = np.array([5000.0, -3000.0, 120.0])
    case = street_case
    res = case["resolutions"]
    om = orc.Map(resolutions=res, default_radius=case["default_radius"])
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=case["default_radius"]))
    for j in range(6):
        pts = case["scans"][j].world_gt + shift
        assert np.array_equal(om.insert(pts), gm.InsertPointCloud(pts))
    sc, raw, t, pose0, _ = _keypoints(case, 6, 0.6)
    pose0 = pose0.copy(); pose0[4:7] += shift; pose0[11:14] += shift
    world0 = se3.ct_transform(pose0, sc.t_begin_end, t, raw)
    o = _opts(num_iters_icp=1, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    s.set_ordering(1 if mode == "rows_ordered" else 0)
    s.set_debug(True)
    s.set_keypoints(raw, world0, t)
    s.solve(pose0, sc.t_begin_end, o)
    dbg = s.get_debug()
    A, b, n_used = s.get_system()
    Ao, bo, no, info = orc.gn_accumulate(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), heap_mode=0, debug=True)
    assert np.array_equal(dbg["n_neighbors"], info["n_neighbors"])
    has = info["n_neighbors"] >= 20
    assert has.sum() > 500 and np.array_equal(dbg["farthest"][has], info["farthest"][has])
    # gate decisions are index work: bit-exact (round 4: the sums, the covariance and the SVD run in the reference build's arithmetic, so
    # the value the |d| < 0.3 gate tests is the oracle's own, however badly conditioned it is 5 km from the origin)
    assert np.array_equal(dbg["used"], info["used"]) and n_used == no
    # default (hybrid) solver: same covariance bits, solver difference ~1e-15 / a2D^2; exact mode: the oracle's bits
    assert np.abs(dbg["a2d"][has] - info["a2d"][has]).max() < 1e-8
    s.set_normals(1)
    s.set_keypoints(raw, world0, t)
    s.solve(pose0, sc.t_begin_end, o)
    dbx = s.get_debug()
    s.set_normals(0)
    assert np.array_equal(dbx["used"], info["used"])
    assert np.array_equal(dbx["a2d"][has], info["a2d"][has]) and np.array_equal(dbx["normal"][has], info["normal"][has])
    assert np.abs(A - Ao).max() < 1e-4 * np.abs(Ao).max()
    o = _opts(num_iters_icp=6, threshold_orientation_norm=0.0)
    s.set_debug(False)
    s.set_keypoints(raw, world0, t)
    pose6, summ6, _ = s.solve(pose0, sc.t_begin_end, o)
    pose_o, _, so = orc.register_gn(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), None, heap_mode=0)
    tr, rot = se3.pose_error(pose6, pose_o)
    assert tr < 1e-6 and rot < 1e-6, (tr, rot)


@pytest.mark.parametrize("mode", ["rows", "rows_ordered"])
def test_k32_neighbours_and_64_point_voxels(box_case, mode):
    """The kernels' limits: max_number_neighbors = 32 (CTGN_MAX_NEIGHBORS) and max_num_points = 64 per voxel."""
    case = dict(box_case, resolutions=[(0.5, 0.02, 64)])
    om, gm = build_maps(case, 6, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(case, 6, 0.4)
    for k, min_nb in ((32, 32), (32, 12), (7, 5)):
        o = _opts(num_iters_icp=1, threshold_orientation_norm=0.0, max_number_neighbors=k, min_number_neighbors=min_nb)
        s = cia.GnSolver(gm)
        s.set_ordering(1 if mode == "rows_ordered" else 0)
        s.set_debug(True)
        s.set_keypoints(raw, world0, t)
        pose1, summ, _ = s.solve(pose0, sc.t_begin_end, o)
        dbg = s.get_debug()
        A, b, n_used = s.get_system()
        Ao, bo, no, info = orc.gn_accumulate(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), heap_mode=0, debug=True)
        assert np.array_equal(dbg["n_neighbors"], info["n_neighbors"]) and info["n_neighbors"].max() == k
        has = info["n_neighbors"] >= max(min_nb, 5)
        assert has.sum() > 1000 and np.array_equal(dbg["farthest"][has], info["farthest"][has])
        assert np.array_equal(dbg["used"], info["used"]) and n_used == no
        assert np.abs(A - Ao).max() < 1e-10 * np.abs(Ao).max()
        pose_o, _, _ = orc.gn_solve_update(Ao, bo, no, None, pose0)
        tr, rot = se3.pose_error(pose1, pose_o)
        assert tr < TIGHT and rot < TIGHT
    got = gm.ComputeNeighborhoods(world0[:300], 32)
    for q, gq in zip(world0[:300], got):
        assert np.array_equal(gq, om.radius_search(q, 0.0, 32, heap_mode=0))


def test_guessed_first_search_bound_changes_nothing_but_the_work(box_case):
    """Round 4: over a level that is dense for its radius the first search of a solve starts from a guess (1.3 x the radius k neighbours
    fill on a surface at the level's points-per-voxel) instead of the radius, and the keypoints the guess leaves with fewer than k
    candidates are searched again on the radius in the same launch (rows_tiles, pass 1). Forced here (ctgn_set_search_guess) with three
    factors — from "most guesses hold" to "most fail and take the second pass": the instrumented instantiation confirms the guess is
    in use, and counts, farthest neighbours, gates, system and poses are identical to the unguessed search's for
    every factor, with pools on and off, after every iteration — and the oracle's on the first accumulation."""
    case = dict(box_case, resolutions=[(0.5, 0.02, 64)])
    om, gm = build_maps(case, 6, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(case, 6, 0.25)
    o = _opts(num_iters_icp=3, threshold_orientation_norm=0.0)
    streamed = {}
    for factor in (0.0, 1.3):
        s = cia.GnSolver(gm)
        s.set_variant(3)
        s.set_search_guess(factor)
        s.set_keypoints(raw, world0, t)
        s.gn_begin(pose0, sc.t_begin_end, o, None)
        s.traffic_counters(reset=True)
        s.gn_iterate(1)
        streamed[factor] = s.traffic_counters(reset=True)[1]
        s.gn_end()
    assert streamed[1.3] != streamed[0.0], streamed          # the guess is in use (how much it saves is the map's business: bench.py, workload D)
    for pools in (0, 1):
        runs = []
        for factor in (0.0, 1.3, 0.7, 0.3):
            s = cia.GnSolver(gm)
            s.set_pools(pools)
            s.set_search_guess(factor)
            s.set_debug(True)
            s.set_keypoints(raw, world0, t)
            s.gn_begin(pose0, sc.t_begin_end, o, None)
            per_iter = []
            for _ in range(o.num_iters_icp):
                s.gn_iterate(1)
                d = s.get_debug()
                per_iter.append((s.get_system(), d["n_neighbors"].copy(), d["farthest"].copy(), d["used"].copy(), d["a2d"].copy()))
            pose, summ, _ = s.gn_end()
            runs.append((pose, per_iter))
        for other in runs[1:]:
            assert np.array_equal(runs[0][0], other[0])
            for k_it, (a, b) in enumerate(zip(runs[0][1], other[1])):
                assert np.array_equal(a[0][0], b[0][0]) and a[0][2] == b[0][2], k_it
                for x, y in zip(a[1:], b[1:]):
                    assert np.array_equal(x, y), k_it
    _, _, _, info = orc.gn_accumulate(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), heap_mode=0, debug=True)
    first = runs[1][1][0]
    assert np.array_equal(first[1], info["n_neighbors"])
    has = info["n_neighbors"] >= 20
    assert has.sum() > 1000 and np.array_equal(first[2][has], info["farthest"][has]) and np.array_equal(first[3], info["used"])


def test_config_d_ouster_scan_matches_oracle():
    """BASELINE.json configs[3] as SURVEY.md section 8d defines it (sharded across GPUs in production; here the single-GPU kernel instantiation
    it uses): an Ouster-128-style scan — 128 beams x 2048 columns x 8 accumulated sub-sweeps = 2.1 M rays RAY-CAST against the residential
    scene — keypoints = the 0.05 m grid of the returns (~0.9 M), the 125-voxel sweep over a 0.5 m x 40-point map (radius 0.8) of every
    surface within 70 m (the bench uses 100 m; 70 keeps the CPU side of this test around half a minute). One accumulation against the
    oracle running OpenMP over keypoints — neighbour counts, farthest neighbours and gate decisions identical, packed system <= 1e-10
    relative — and then the driving profile's FIVE iterations: same iteration count, same residual count, pose <= 1e-8. Same generator
    as `bench.py --workload D`."""
    import os
    import bench
    inp = bench.make_inputs_ouster(sweeps=8, radius=70.0)
    assert inp["rays"] == 128 * 2048 * 8
    res = [(0.5, 0.03, 40)]
    om = orc.Map(resolutions=res, default_radius=0.8)
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*res[0])], default_radius=0.8))
    pts = inp["map_points"]
    for s0 in range(0, len(pts), 2_000_000):
        kept = gm.InsertPointCloud(pts[s0:s0 + 2_000_000])
        assert np.array_equal(kept, om.insert(pts[s0:s0 + 2_000_000]))
    assert gm.SearchParamsFromRadiusSearch() == (0, 0.5, 2)
    raw, t = inp["raw"], inp["t"]
    n = len(t)
    assert n >= 800_000
    pose0 = syn.perturb_pose(inp["pose_gt"], 0.003, 0.02, seed=5)
    world0 = se3.ct_transform(pose0, inp["tbe"], t, raw)
    threads = max(1, min(16, os.cpu_count() or 1))
    o = _opts(num_iters_icp=1, threshold_orientation_norm=0.0)
    Ao, bo, no, info = orc.gn_accumulate(om, raw, world0, t, pose0, inp["tbe"], _oopts(o), heap_mode=0, num_threads=threads, debug=True)
    o5 = _opts(num_iters_icp=5, threshold_orientation_norm=0.0)
    pose_o5, _, so5 = orc.register_gn(om, raw, world0, t, pose0, inp["tbe"], _oopts(o5), None, heap_mode=0, num_threads=threads)
    for ordering in (0, 1):                                        # caller order; home-voxel order (what the library picks for this size)
        s = cia.GnSolver(gm)
        s.set_ordering(ordering)
        s.set_debug(True)
        s.set_keypoints(raw, world0, t)
        s.solve(pose0, inp["tbe"], o)
        dbg = s.get_debug()
        A, b, n_used = s.get_system()
        assert np.array_equal(dbg["n_neighbors"], info["n_neighbors"])
        has = info["n_neighbors"] >= 20
        assert has.sum() > 500_000 and np.array_equal(dbg["farthest"][has], info["farthest"][has])
        assert np.array_equal(dbg["used"], info["used"]) and n_used == no
        assert np.abs(A - Ao).max() < 1e-10 * np.abs(Ao).max() and np.abs(b - bo).max() < 1e-10 * np.abs(bo).max() + 1e-14
        s.set_debug(False)
        s.set_keypoints(raw, world0, t)
        pose5, summ5, _ = s.solve(pose0, inp["tbe"], o5)
        assert summ5.num_iters == so5.num_iters == 5 and summ5.num_residuals_used == so5.num_residuals_used
        tr, rot = se3.pose_error(pose5, pose_o5)
        assert tr < TIGHT and rot < TIGHT, (tr, rot)
        # iterations 3-5 ran on neighbour pools (99.9 % of the keypoints certified on this scan): searching every iteration gives the same bits
        w5 = s.world_points()
        s.set_pools(0)
        s.set_keypoints(raw, world0, t)
        pose5_np, summ5_np, _ = s.solve(pose0, inp["tbe"], o5)
        assert np.array_equal(pose5, pose5_np) and np.array_equal(w5, s.world_points()) and summ5_np.num_residuals_used == summ5.num_residuals_used


# ------------------------------------------------------------------------------------------------- two GPUs (skipped on a 1-GPU box)
def _two_gpus():
    import torch
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


def _sharded_worker(rank, world, port, tmpdir):
    import os, sys
    sys.path.insert(0, ROOT_DIR); sys.path.insert(0, os.path.join(ROOT_DIR, "tests"))
    import torch
    import torch.distributed as dist
    import conftest
    from ct_icp_amd.distributed import ShardedGnSolver, home_voxel_order, shard_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    case = conftest.street_case.__wrapped__()
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in case["resolutions"]],
                                                default_radius=case["default_radius"], device=rank))
    for j in range(6):
        gm.InsertPointCloud(case["scans"][j].world_gt)                 # the map is replicated
    sc, raw, t, pose0, world0 = _keypoints(case, 6, 0.3)
    order = home_voxel_order(world0, case["resolutions"][0][0])        # global home-voxel sort, then contiguous chunks
    lo, hi = shard_bounds(len(t), world, rank)
    idx = order[lo:hi]
    sh = ShardedGnSolver(gm)
    sh.set_keypoints(raw[idx], world0[idx], t[idx])
    pose, summ, _ = sh.solve(pose0, sc.t_begin_end, _opts(num_iters_icp=5, threshold_orientation_norm=0.0))
    np.save(os.path.join(tmpdir, f"pose_{rank}.npy"), np.concatenate([pose, [summ.num_residuals_used, summ.num_iters]]))
    sh.close()
    dist.barrier()
    dist.destroy_process_group()


ROOT_DIR = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_two_gpu_shard_systems_sum_to_the_single_gpu_system(street_case):
    """SURVEY.md section 8e on hardware: the packed systems of two GPU shards (contiguous chunks of the home-voxel-sorted keypoints,
    map replicated on both devices) add up to the single-GPU system (<= 1e-12 relative; same count)."""
    from ct_icp_amd.distributed import home_voxel_order, shard_bounds
    case = street_case
    maps = []
    for dev in (0, 1):
        gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in case["resolutions"]],
                                                    default_radius=case["default_radius"], device=dev))
        for j in range(6):
            gm.InsertPointCloud(case["scans"][j].world_gt)
        maps.append(gm)
    sc, raw, t, pose0, world0 = _keypoints(case, 6, 0.3)
    o = _opts(num_iters_icp=1, threshold_orientation_norm=0.0)
    order = home_voxel_order(world0, case["resolutions"][0][0])
    s = cia.GnSolver(maps[0])
    s.set_keypoints(raw[order], world0[order], t[order])
    s.solve(pose0, sc.t_begin_end, o)
    A, b, n = s.get_system()
    As, bs, ns = np.zeros_like(A), np.zeros_like(b), 0
    for dev in (0, 1):
        lo, hi = shard_bounds(len(t), 2, dev)
        idx = order[lo:hi]
        sd = cia.GnSolver(maps[dev])
        sd.set_keypoints(raw[idx], world0[idx], t[idx])
        sd.gn_begin(pose0, sc.t_begin_end, o, None)
        sd.gn_accumulate()
        Ad, bd, nd = sd.get_system()
        sd.gn_solve_update(); sd.gn_end()
        As += Ad * 1.0; bs += bd; ns += nd
    assert ns == n and np.abs(As - A).max() < 1e-12 * np.abs(A).max() and np.abs(bs - b).max() < 1e-12 * np.abs(b).max() + 1e-18


@pytest.mark.skipif(not _two_gpus(), reason="needs two GPUs")
def test_two_rank_sharded_registration_over_rccl(street_case, tmp_path):
    """Two processes, one GPU each, ncclAllReduce issued by the library: both ranks end on the identical pose, equal (<= 1e-12) to the
    single-GPU registration of the whole keypoint set."""
    import torch.multiprocessing as mp
    mp.spawn(_sharded_worker, args=(2, 29541, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "pose_0.npy"), np.load(tmp_path / "pose_1.npy")
    assert np.array_equal(p0, p1)
    om, gm = build_maps(street_case, 6, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(street_case, 6, 0.3)
    s = cia.GnSolver(gm)
    s.set_keypoints(raw, world0, t)
    pose, summ, _ = s.solve(pose0, sc.t_begin_end, _opts(num_iters_icp=5, threshold_orientation_norm=0.0))
    assert int(p0[14]) == summ.num_residuals_used and int(p0[15]) == summ.num_iters
    assert np.abs(p0[:14] - pose).max() < 1e-12


@pytest.mark.parametrize("ordering", [0, 1])
def test_soft_failure_at_a_later_iteration_returns_the_last_completed_state(box_case, ordering):
    """ct_icp.cpp:860-871 at iteration i >= 1: the poses are those after iteration i - 1 and the world points the ones that iteration
    wrote (:964-966). A constant-velocity prior with an absurd previous velocity throws the first update metres away, so the second
    iteration finds fewer than 100 usable keypoints. Same answer in caller order, in home-voxel order (where the GN kernels iterate on
    a position-ordered working copy) and from the oracle; the device-resident map path rides along."""
    om, gm = build_maps(box_case, 5, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(box_case, 5, 0.5)
    mm = cia.PreviousFrameMotionModel(beta_location_consistency=1e6, beta_constant_velocity=1e6)
    far = np.zeros(14); far[3] = far[10] = 1.0; far[11:14] = [40.0, 0.0, 0.0]
    mm.previous_frame = cia.TrajectoryFrame.from_pose14(far, 0.0, 0.0)
    op = orc.MotionPrior(beta_location_consistency=1e6, beta_constant_velocity=1e6, previous_begin_tr=far[4:7], previous_end_tr=far[11:14])
    o = _opts(num_iters_icp=6, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    s.set_ordering(ordering)
    s.set_keypoints(raw, world0, t)
    pose1, summ, _ = s.solve(pose0, sc.t_begin_end, o, mm)
    pose_o, world_o, so = orc.register_gn(om, raw, world0, t, pose0, sc.t_begin_end, _oopts(o), op, heap_mode=0)
    assert not so.success and so.num_iters == 2                              # the oracle fails in its third iteration, after two completed ones
    assert not summ.success and summ.num_iters == so.num_iters and summ.num_residuals_used == so.num_residuals_used
    assert summ.error_log == so.error_log
    tr, rot = se3.pose_error(pose1, pose_o)
    assert tr < 1e-5 and rot < 1e-6                                          # a 40 m jump through a 1e6-weighted prior: conditioning, not parity
    assert np.abs(s.world_points() - world_o).max() < 1e-4
    assert np.abs(s.world_points() - world0).max() > 10.0                    # not the uploaded points: the last completed iteration's


# ------------------------------------------------------------------------------------------------- round 3: rewind, library-side sharding
def test_rewind_restarts_a_solve_from_the_uploaded_world_points(street_case):
    """ctgn_set_rewind / ctgn_rewind_keypoints (include/ctgn.h): a registration repeated on the resident keypoints from their uploaded world
    points — the retry loop around TryRegister, reference src/ct_icp/odometry.cpp:794-845 — without another upload. Bit-identical to a
    fresh upload + solve; without the rewind the second solve starts from the first one's world points and differs."""
    om, gm = build_maps(street_case, 6, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(street_case, 6, 0.4)
    o = _opts(num_iters_icp=4, threshold_orientation_norm=0.0)
    mm, _ = _prior(street_case, 6)
    s = cia.GnSolver(gm)
    with pytest.raises(L.CtgnError):
        s.rewind()                                          # nothing saved yet
    s.set_rewind(True)
    s.set_keypoints(raw, world0, t)
    pose_a, summ_a, _ = s.solve(pose0, sc.t_begin_end, o, mm)
    world_a = s.world_points()
    pose_n, _, _ = s.solve(pose0, sc.t_begin_end, o, mm)                 # no rewind: iteration 0 searches at the previous solve's world points
    assert np.abs(pose_n - pose_a).max() > 0
    s.rewind()
    pose_b, summ_b, _ = s.solve(pose0, sc.t_begin_end, o, mm)
    assert np.array_equal(pose_a, pose_b) and summ_a.num_residuals_used == summ_b.num_residuals_used
    assert np.array_equal(world_a, s.world_points())
    # back-to-back solves without ending the previous one (what bench.py times): the last one still equals a fresh solve
    s.set_profiling(True)
    s.kernel_timing(reset=True)
    for _ in range(3):
        s.rewind()
        s.gn_begin(pose0, sc.t_begin_end, o, mm)
        s.gn_iterate(4)
    pose_c, summ_c, _ = s.gn_end()
    assert np.array_equal(pose_a, pose_c) and summ_c.num_iters == 4
    (ms0, n0), (ms1, n1) = s.kernel_timing_split(reset=True)
    assert n0 == 3 and n1 == 9 and ms0 > 0 and ms1 > 0     # 3 first-of-solve searches (radius only), 9 with a carried-over bound
    s.set_profiling(False)


def test_library_side_sharding_on_one_gpu(street_case):
    """ctgn_set_keypoints_sharded: every rank hands over the whole scan, the library sorts it by home voxel and keeps
    the rank's contiguous chunk. Two 'ranks' played one after the other on the one GPU: the chunks partition the scan, each chunk is a
    contiguous run of the voxel-sorted order, the two packed systems add up to the unsharded system, and a timestamp outside the pose
    interval anywhere in the scan is refused by EVERY rank (no rank left waiting in the all-reduce)."""
    om, gm = build_maps(street_case, 6, with_gpu=True)
    sc, raw, t, pose0, world0 = _keypoints(street_case, 6, 0.3)
    o = _opts(num_iters_icp=1, threshold_orientation_norm=0.0)
    s = cia.GnSolver(gm)
    s.set_keypoints(raw, world0, t)
    s.solve(pose0, sc.t_begin_end, o)
    A, b, n = s.get_system()
    As, bs, ns, seen = np.zeros_like(A), np.zeros_like(b), 0, []
    res = street_case["resolutions"][0][0]
    for rank in (0, 1):
        idx = s.set_keypoints_sharded(raw, world0, t, rank, 2)
        seen.append(idx)
        # round 5: a rank uploads the scan's world points (the order every rank must agree on) + its own chunk of the seven arrays, not the scan
        pad = lambda v: (v + 63) // 64 * 64
        assert s.last_upload_bytes() == 8 * (3 * pad(len(t)) + 7 * pad(len(idx))) < 8 * 7 * len(t) * 0.95
        vox = np.trunc(world0[idx] / res).astype(np.int64)
        key = ((vox[:, 0] & 4095) << 20) | ((vox[:, 1] & 4095) << 8) | (vox[:, 2] & 255)
        assert np.all(np.diff(key) >= 0)                    # resident order = home-voxel order
        s.gn_begin(pose0, sc.t_begin_end, o, None)
        s.gn_accumulate()
        Ad, bd, nd = s.get_system()
        s.gn_solve_update(); s.gn_end()
        As += Ad; bs += bd; ns += nd
        w = s.world_points()
        assert w.shape == (len(idx), 3)
    both = np.concatenate(seen)
    assert len(both) == len(t) and np.array_equal(np.sort(both), np.arange(len(t))) and abs(len(seen[0]) - len(seen[1])) <= 1
    assert ns == n and np.abs(As - A).max() < 1e-12 * np.abs(A).max() and np.abs(bs - b).max() < 1e-12 * np.abs(b).max() + 1e-18
    # one bad timestamp in the OTHER rank's half: this rank refuses the solve too
    t_bad = t.copy()
    t_bad[seen[1][0]] = sc.t_begin_end[1] + 1.0
    s.set_keypoints_sharded(raw, world0, t_bad, 0, 2)
    with pytest.raises(L.CtgnError) as e:
        s.gn_begin(pose0, sc.t_begin_end, o, None)
    assert e.value.status == L.ERR_TIMESTAMP_RANGE


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_bench_ranks_on_one_gpu_gloo_rehearsal(ranks):
    """The N > 1 path of bench.py where the driver's GPU tier can see it: 2, 4 and 8 gloo ranks sharing the one GPU run config D (reduced: one
    sub-sweep of the Ouster pattern) through ctgn_set_keypoints_sharded and the sharded loop — the shard bounds of every world size the
    driver's scaling run uses, the exchange of the packed system among that many ranks, and the fail-together bookkeeping of the loop; the
    line must carry the strong-scaling fields and the pose parity of the sharded solve against the oracle on the whole scan."""
    import json, os, subprocess, sys
    root = ROOT_DIR
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # plain `python bench.py --gpus N`: the script starts its own N ranks (round 5; the driver's N > 1 command line goes through the same code)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--dist-backend", "gloo", "--steps", "5", "--warmup", "0", "--clock-warm", "0",
           "--workload", "D", "--d-sweeps", "1", "--d-radius", "45" if ranks == 2 else "30", "--no-pmc"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()           # (torch's gloo backend announces its ranks on stdout; the line the driver parses is the LAST one)
    assert len(lines[-1]) < 4096 and sum(ln.startswith("{") for ln in lines) == 1, [len(ln) for ln in lines]
    d = json.loads(lines[-1])
    full = json.load(open(os.path.join(root, "bench_detail.json")))                 # everything else: the detail file
    assert full["value"] == d["value"]
    d["parity"] = full["parity"]
    assert d["n_gpus"] == ranks and d["scaling"] == "strong" and d["config"]["workload_id"] == "D"
    assert d["config"]["keypoints_total"] > 50_000 and abs(d["config"]["keypoints_per_gpu"] * ranks - d["config"]["keypoints_total"]) < ranks
    assert d["parity_m_rad"][0] < 1e-4 and d["parity_m_rad"][1] < 1e-4 and d["parity"]["n_used_gpu"] == d["parity"]["n_used_oracle"]
    assert d["strong_scaling_single_gpu_reference"]["keypoints"] == d["config"]["keypoints_total"] and "weak_scaling_line" in d


@pytest.mark.parametrize("key_bytes", [8, 4])
def test_hand_written_radix_sort_and_compaction(box_case, key_bytes):
    """ctgn_sort.hpp (no sort library underneath since round 3): stable LSD radix sort of (key, index) pairs — one launch up to 16 k keys,
    three kernels per executed pass beyond — against NumPy's stable argsort, on sizes around every boundary (round of 64, the one-block
    limit, a column tile), keys whose bytes partly never vary (skipped passes) or never vary at all; ordered compaction against
    np.flatnonzero."""
    import ctypes as C
    om, gm = build_maps(box_case, 1, with_gpu=True)
    h = gm.handle
    rng = np.random.default_rng(5)
    bits = 64 if key_bytes == 8 else 32
    for n in (1, 2, 63, 64, 65, 1000, 7731, 16384, 16385, 70_000, 1_300_000):
        for kind in ("random", "few", "constant", "voxel"):
            if kind == "random":
                keys = rng.integers(0, 2 ** bits, n, dtype=np.uint64)
            elif kind == "few":                                     # many duplicates: stability is what is tested
                keys = rng.integers(0, 37, n, dtype=np.uint64) << np.uint64(8 if key_bytes == 4 else 40)
            elif kind == "constant":
                keys = np.full(n, 0x0123456789ABCDEF & (2 ** bits - 1), dtype=np.uint64)
            else:                                                   # packed voxel keys of a frame: three 21-bit biased coordinates
                v = rng.integers(-120, 121, (n, 3)).astype(np.int64) + (1 << 20)
                keys = (v[:, 0] | (v[:, 1] << 21) | (v[:, 2] << 42)).astype(np.uint64) & np.uint64(2 ** bits - 1)
            order = np.zeros(n, dtype=np.uint32)
            L.check(h, L.lib().ctgn_test_sort_pairs(h, keys.ctypes.data_as(C.POINTER(C.c_uint64)), n, bits, key_bytes,
                                                   order.ctypes.data_as(C.POINTER(C.c_uint32))))
            assert np.array_equal(order, np.argsort(keys, kind="stable").astype(np.uint32)), (n, kind)
        flags = (rng.random(n) < 0.3).astype(np.uint8)
        out = np.zeros(n, dtype=np.uint32)
        cnt = C.c_size_t()
        L.check(h, L.lib().ctgn_test_compact(h, flags.ctypes.data_as(C.POINTER(C.c_uint8)), n, out.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cnt)))
        assert np.array_equal(out[:cnt.value], np.flatnonzero(flags).astype(np.uint32)), n
