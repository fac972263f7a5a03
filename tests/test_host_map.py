"""CPU tests of the product's host-side map mirror (ct_icp_amd/csrc/ctgn_map.hpp through the C ABI, device = -1)
against the oracle map: same insert decisions, same point sets, same eviction, same search parameters."""
import numpy as np

import ct_icp_amd as cia
from oracle import oracle as orc


def _sorted(a):
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def _pair(res, radius):
    gm = cia.GpuVoxelMap(cia.GpuVoxelMapOptions(resolutions=[cia.ResolutionParam(*r) for r in res], default_radius=radius,
                                                device=-1))
    om = orc.Map(resolutions=res, default_radius=radius)
    return gm, om


def test_insert_rule_matches_oracle_on_every_resolution(box_case):
    res = [(0.2, 0.03, 50), (0.5, 0.1, 40), (1.5, 0.15, 40)]
    gm, om = _pair(res, 0.8)
    for j in range(3):
        pts = box_case["scans"][j].world_gt
        kg, ko = gm.InsertPointCloud(pts), om.insert(pts)
        assert np.array_equal(kg, ko)
    assert gm.NumPoints() == om.num_points()
    for li in range(3):
        assert gm.NumVoxels(li) == om.num_voxels(li)
        assert np.array_equal(_sorted(gm.MapAsPointCloud(li)), _sorted(om.export(li)))
    assert gm.SearchParamsFromRadiusSearch() == om.search_params()
    assert gm.SearchParamsFromRadiusSearch(1.7) == om.search_params(1.7)


def test_float32_strided_input_is_cast_like_the_proxy_view(box_case):
    gm, om = _pair([(0.5, 0.05, 20)], 0.8)
    pts = box_case["scans"][0].world_gt[:4000]
    rec = np.zeros(len(pts), dtype=[("pad", "<f4"), ("xyz", "<f4", 3), ("i", "<u2")])
    rec["xyz"] = pts
    view = rec["xyz"]                              # float32, strided
    gm.InsertPointCloud(view)
    om.insert(view.astype(np.float64))
    assert np.array_equal(_sorted(gm.MapAsPointCloud(0)), _sorted(om.export(0)))


def test_eviction_tombstones_and_reuse(street_case):
    gm, om = _pair(street_case["resolutions"], street_case["default_radius"])
    rng = np.random.default_rng(0)
    for j in range(8):
        sc = street_case["scans"][j]
        pts = sc.world_gt[rng.permutation(len(sc.world_gt))[:15000]]
        gm.InsertPointCloud(pts)
        om.insert(pts)
        loc = sc.pose_gt[11:14]
        gm.RemoveElementsFarFromLocation(loc, 25.0)       # aggressive: forces tombstones, block reuse and rehashes
        om.remove_far(loc, 25.0)
        assert gm.NumPoints() == om.num_points() and gm.NumVoxels(0) == om.num_voxels(0)
    assert np.array_equal(_sorted(gm.MapAsPointCloud(0)), _sorted(om.export(0)))
    gm.ClearMap()
    assert gm.NumPoints() == 0 and gm.NumVoxels(0) == 0
    gm.InsertPointCloud(street_case["scans"][0].world_gt[:100])
    assert gm.NumPoints() > 0


def test_out_of_range_and_non_finite_points_are_skipped_not_fatal(box_case):
    """ctgn_map_insert on a batch that contains points outside the 21-bit voxel key range (or NaN / inf): the in-range points are
    inserted, the others report inserted = 0, the call returns CTGN_OK (a retry after an error would have inserted the batch twice)
    and the map equals the oracle's map of the in-range points."""
    gm, om = _pair([(0.5, 0.05, 20)], 0.8)
    pts = box_case["scans"][0].world_gt[:3000].copy()
    bad = np.array([10, 500, 1234, 2999])
    ok = np.ones(len(pts), dtype=bool); ok[bad] = False
    pts[10] = [3e9, 0.0, 0.0]
    pts[500] = [np.nan, 1.0, 1.0]
    pts[1234] = [0.0, -np.inf, 0.0]
    pts[2999] = [0.0, 0.0, -7e8]
    kept = gm.InsertPointCloud(pts)                      # must not raise
    assert not kept[bad].any()
    assert np.array_equal(kept[ok], om.insert(pts[ok]))
    assert gm.NumPoints() == om.num_points()
    assert np.array_equal(_sorted(gm.MapAsPointCloud(0)), _sorted(om.export(0)))
