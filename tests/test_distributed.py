"""world_size 2 / 4 / 8 tests of the keypoint-sharded mode's exchange step on CPU (gloo backend).

On GPUs each rank runs libctgn's accumulate kernel on its shard and the packed system (96 doubles) is all-reduced over
RCCL (ct_icp_amd/distributed.py). Here the per-shard systems come from the CPU oracle (the checker), so the test pins the
sharding, the packing layout and the collective: sum over ranks of the shard systems == the system of the whole keypoint
set, and every rank derives the identical pose update from it."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from ct_icp_amd import se3, synthetic as syn
    from ct_icp_amd.distributed import allreduce_system, pack_system, shard_bounds, unpack_system
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # identical, seeded inputs on every rank (the map is replicated)
    scene = syn.box_scene(6.0, n_spheres=2, seed=7)
    el = np.radians(np.linspace(-70, 70, 30)); az = np.linspace(0, 2 * np.pi, 160, endpoint=False)
    dirs = np.stack([np.outer(np.cos(az), np.cos(el)), np.outer(np.sin(az), np.cos(el)), np.outer(np.ones_like(az), np.sin(el))], -1).reshape(-1, 3)
    rel_t = np.repeat(np.arange(len(az)) / len(az), len(el))
    knots = np.zeros((4, 7)); knots[:, 3] = 1.0
    for j in range(4):
        knots[j, 4:] = [0.1 * j, 0.05 * j, 0.0]
    om = orc.Map(resolutions=[(0.5, 0.05, 20)], default_radius=0.8)
    for j in range(2):
        om.insert(syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, j), 0.1 * j, 0.1 * (j + 1), 30.0, 0.3, 0.01, j).world_gt)
    sc = syn.generate_scan(scene, dirs, rel_t, syn.frame_pose14(knots, 2), 0.2, 0.3, 30.0, 0.3, 0.01, 9)
    pose0 = syn.perturb_pose(sc.pose_gt, 0.005, 0.03, seed=1)
    world0 = se3.ct_transform(pose0, sc.t_begin_end, sc.t, sc.raw)
    opts = orc.Options(num_iters_icp=1)
    n = len(sc.t)
    lo, hi = shard_bounds(n, world, rank)
    A, b, nu = orc.gn_accumulate(om, sc.raw[lo:hi], world0[lo:hi], sc.t[lo:hi], pose0, sc.t_begin_end, opts)
    system = torch.from_numpy(pack_system(A, b, nu))
    allreduce_system(system)                                   # the one collective of the path
    Ar, br, nr = unpack_system(system.numpy())
    Af, bf, nf = orc.gn_accumulate(om, sc.raw, world0, sc.t, pose0, sc.t_begin_end, opts)
    ok = (nr == nf) and np.allclose(Ar, Af, rtol=1e-12, atol=1e-12) and np.allclose(br, bf, rtol=1e-12, atol=1e-12)
    pose1, x, _ = orc.gn_solve_update(Ar, br, nr, None, pose0)
    gathered = [torch.zeros(14, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(pose1))
    same = all(torch.equal(g, gathered[0]) for g in gathered)   # identical reduced input -> identical update, no broadcast needed
    np.save(os.path.join(tmpdir, f"ok_{rank}.npy"), np.array([ok, same, hi - lo, nr]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from ct_icp_amd.distributed import shard_bounds
    for n in (0, 1, 7, 100, 132339):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    from ct_icp_amd.distributed import pack_system, unpack_system
    rng = np.random.default_rng(0)
    J = rng.normal(size=(30, 12))
    A, b = J.T @ J, rng.normal(size=12)
    A2, b2, n2 = unpack_system(pack_system(A, b, 17))
    assert np.array_equal(A, A2) or np.allclose(A, A2, rtol=0, atol=0)
    assert np.array_equal(b, b2) and n2 == 17


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_allreduce_of_packed_system(tmp_path, world):
    """World sizes 2, 4 and 8 — the ones the driver's scaling run uses: the shards cover the scan, their packed systems sum to the whole
    scan's, every rank derives the same pose from the reduced system."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"ok_{r}.npy") for r in range(world)]
    assert all(r[0] == 1 and r[1] == 1 for r in res), res
    assert len({int(r[3]) for r in res}) == 1 and res[0][3] > 100
    sizes = [int(r[2]) for r in res]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) > 1000


# ------------------------------------------------------------------------------------------------- config E: sequences per rank
def _batch_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ct_icp_amd import sequence_runner as sr
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = [40, 10, 41, 8, 3, 25, 10]
    ran = []

    def fake_run_sequence(scans, **kw):                       # stands in for the GPU loop: one millisecond per frame
        import time
        import torch
        assert kw["device"] == rank % max(1, torch.cuda.device_count())       # run_batch maps the rank to its own device
        time.sleep(1e-3 * len(scans))
        ran.append(len(scans))
        return dict(poses=np.zeros((len(scans), 14)), success=np.ones(len(scans), bool), seconds=float(len(scans)), frames=len(scans),
                    keypoints=np.zeros(len(scans)), sampled=np.zeros(len(scans)), map_points=0)

    sr.run_sequence = fake_run_sequence
    _, shares = sr.deal_sequences(lengths, world)
    mine = {sid: [None] * lengths[sid] for sid in shares[rank]}          # a rank only holds its own sequences
    out = sr.run_batch(mine, lengths, rank=rank, world_size=world)
    np.save(os.path.join(tmpdir, f"batch_{rank}.npy"), np.array([out["frames"], out["wall_seconds"], sum(ran), len(out["results"])]))
    dist.barrier()
    dist.destroy_process_group()


def test_config_e_sequences_are_dealt_longest_first_and_aggregated(tmp_path):
    """SURVEY.md 8d config E on CPU (gloo, world size 2): every rank runs only its own sequences (longest first, round-robin), nothing
    is exchanged while they run, the gathered result counts every frame once and the job's wall time is the slowest rank's."""
    import torch.multiprocessing as mp
    from ct_icp_amd import sequence_runner as sr
    lengths = [40, 10, 41, 8, 3, 25, 10]
    order, shares = sr.deal_sequences(lengths, 2)
    assert order == [2, 0, 5, 1, 6, 3, 4] and shares == [[2, 5, 6, 4], [0, 1, 3]]
    assert sorted(shares[0] + shares[1]) == list(range(7))
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_batch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "batch_0.npy"), np.load(tmp_path / "batch_1.npy")
    assert r0[0] == r1[0] == sum(lengths) and r0[3] == r1[3] == 7
    assert r0[2] == 41 + 25 + 10 + 3 and r1[2] == 40 + 10 + 8
    # the job's wall time is the slowest rank's measured wall time over its whole share (1 ms per frame in the stand-in)
    assert r0[1] == r1[1] and 1e-3 * max(41 + 25 + 10 + 3, 40 + 10 + 8) <= r0[1] < 5.0
