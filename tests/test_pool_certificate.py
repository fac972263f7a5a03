"""The neighbour-pool certificate of the row kernel (ct_icp_amd/csrc/ctgn_kernels.hpp phase V), restated in
NumPy and checked against brute force: whenever the certificate passes, the k nearest pool members ARE the k nearest map points
within the radius, in the same order — for pools left by bounded searches (pool capped, pool complete), for chains of pool checks
(the completeness radius shrinking by every move), for sparse neighbourhoods (fewer than k points within the radius) and with the
radius stored the way the kernel stores it (float32, rounded down). CPU only: this pins the ARGUMENT; the kernel's use of it is pinned
by the bit-identity tests of the GPU tier (pools on / off)."""
import numpy as np
import pytest

K, POOL_CAP = 20, 28


def f32_down(x):
    y = np.float32(x)
    return y if float(y) <= x else np.nextafter(y, np.float32(-np.inf))


def f32_up(x):
    y = np.float32(x)
    return y if float(y) >= x else np.nextafter(y, np.float32(np.inf))


def bounded_search(points, p, bound2, radius2):
    """What a bounded search leaves: the candidates within min(bound, radius), sorted; the pool = its first POOL_CAP; the radius inside
    which the pool is complete = sqrt(min(admission bound, distance of the last pool member if something was dropped)), rounded down."""
    d2 = ((points - p) ** 2).sum(1)
    adm = min(bound2, radius2)
    idx = np.nonzero(d2 <= adm)[0]
    idx = idx[np.argsort(d2[idx], kind="stable")]
    pool = idx[:POOL_CAP]
    r2 = adm if len(idx) <= POOL_CAP else min(adm, d2[pool[-1]])
    R = f32_down(np.sqrt(r2) * (1.0 - 1e-12))
    n = min(K, len(idx))
    kth = f32_up(np.sqrt(d2[pool[K - 1]] if n >= K else radius2) * (1.0 + 1e-12))
    return pool, float(R), float(kth)


def pool_check(points, pool, R_prev, p_prev, p, radius2):
    """Phase V for one keypoint: (passed, neighbours in order, new R)."""
    moved = float(np.sqrt(((p - p_prev) ** 2).sum()))
    rnow = R_prev - moved * (1.0 + 1e-9) - 1e-12
    if rnow <= 0.0:
        return False, None, 0.0
    rr2 = float(f32_down(rnow * rnow * (1.0 - 1e-9)))
    d2 = ((points[pool] - p) ** 2).sum(1)
    order = np.argsort(d2, kind="stable")
    inside = d2[order] <= radius2
    n_in = int(inside.sum())
    need2 = d2[order[K - 1]] if n_in >= K else radius2
    if not (need2 * (1.0 + 1e-8) < rr2):
        return False, None, 0.0
    n = min(n_in, K)
    s = np.float32(np.sqrt(np.float32(rr2)))
    R_new = float(np.nextafter(np.nextafter(s, np.float32(-np.inf)), np.float32(-np.inf)))
    return True, pool[order[:n]], R_new


def brute(points, p, radius2):
    d2 = ((points - p) ** 2).sum(1)
    idx = np.nonzero(d2 <= radius2)[0]
    return idx[np.argsort(d2[idx], kind="stable")][:K]


@pytest.mark.parametrize("density, radius", [(4000, 0.75), (600, 0.75), (150, 0.8), (20000, 0.5)])
def test_certified_pools_equal_brute_force(density, radius):
    rng = np.random.default_rng(density)
    radius2 = radius * radius
    passed = failed = sparse_passed = 0
    for trial in range(300):
        pts = rng.uniform(-1.5, 1.5, size=(density, 3))
        if trial % 3 == 0:                                   # a surface-like cloud: what a scan map looks like
            pts[:, 2] = 0.02 * rng.standard_normal(density)
        p0 = rng.uniform(-0.3, 0.3, size=3)
        d2 = np.sort(((pts - p0) ** 2).sum(1))
        kth0 = np.sqrt(d2[K - 1]) if d2[K - 1] <= radius2 else radius
        moved0 = abs(rng.normal(0.0, 0.1 * kth0))             # moves in proportion to the neighbourhood's size
        pool, R, _ = bounded_search(pts, p0, (kth0 + moved0) ** 2 * (1.0 + 1e-9), radius2)      # the kernel's bound: previous k-th + move
        p_prev = p0
        for step in range(4):                                # a chain of pool checks: moves getting smaller, R shrinking with each
            p = p_prev + rng.normal(0.0, 0.05 * kth0 / (1 + step), size=3)
            ok, nbrs, R_new = pool_check(pts, pool, R, p_prev, p, radius2)
            if not ok:
                failed += 1
                break
            want = brute(pts, p, radius2)
            assert np.array_equal(nbrs, want), (trial, step)
            passed += 1
            sparse_passed += len(want) < K
            R, p_prev = R_new, p
    assert passed > 50, (passed, failed)                     # the certificate is not vacuous on these clouds ...
    assert failed > 0 or density >= 20000, (passed, failed)  # ... and it does refuse when the margin is gone


def test_a_point_just_outside_the_pool_is_never_missed():
    """Adversarial: a map point placed right at the pool's completeness radius, the keypoint moved straight towards it. The check must
    refuse as soon as that point could be among the k nearest."""
    rng = np.random.default_rng(7)
    radius2 = 0.75 ** 2
    refused = accepted = 0
    for trial in range(400):
        pts = rng.uniform(-1.0, 1.0, size=(3000, 3))
        p0 = np.zeros(3)
        d = np.sqrt(np.sort((pts ** 2).sum(1)))
        pool, R, _ = bounded_search(pts, p0, (d[K - 1] * 1.05) ** 2, radius2)
        direction = rng.standard_normal(3); direction /= np.linalg.norm(direction)
        lurker = direction * (R * (1.0 + 1e-6))               # not in the pool (beyond R), as close as a non-member can be
        pts2 = np.vstack([pts, lurker])
        step = rng.uniform(0.0, 0.06)
        p = direction * step
        ok, nbrs, _ = pool_check(pts2, pool, R, p0, p, radius2)
        want = brute(pts2, p, radius2)
        if ok:
            accepted += 1
            assert np.array_equal(nbrs, want), trial
        else:
            refused += 1
    assert refused > 0 and accepted > 0, (refused, accepted)

