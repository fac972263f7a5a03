"""Keypoint-sharded multi-GPU GN: the map is replicated, every rank owns a contiguous shard
of the keypoints, and the only exchange per GN iteration is one all-reduce (sum) of the packed normal equations —
96 doubles: 78 upper-triangular JtJ | 12 Jtr | count | pad — over RCCL/xGMI (`torch.distributed` backend "nccl").
Every rank then runs the identical 12x12 solve on identical input, so no pose broadcast is needed.

The reference has no collective to mirror (it is a single process); this is the one the path needs
(ct_icp.cpp:843-850 sums over keypoints; :877-882 needs the global count).
"""
from __future__ import annotations

import numpy as np

SYSTEM_DOUBLES = 96


def shard_bounds(n: int, world_size: int, rank: int):
    """Contiguous, balanced [lo, hi) shard of n keypoints (first n % world_size ranks get one more)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def home_voxel_order(world_points: np.ndarray, resolution: float) -> np.ndarray:
    """Permutation that sorts keypoints by the home voxel of their world point (x, then y, then z voxel index; stable). Sharding the
    SORTED sequence into contiguous chunks (shard_bounds) gives every GPU a compact region of the replicated map to touch
    ("keypoints split into G contiguous chunks after the voxel-key sort")."""
    v = np.trunc(np.asarray(world_points, dtype=np.float64) / resolution).astype(np.int64)
    return np.lexsort((v[:, 2], v[:, 1], v[:, 0]))


def pack_system(A: np.ndarray, b: np.ndarray, n_used: int) -> np.ndarray:
    """Host packing of (A, b, count) in the device layout — used by the CPU (gloo) tests of the exchange."""
    s = np.zeros(SYSTEM_DOUBLES)
    s[:78] = np.asarray(A)[np.triu_indices(12)]
    s[78:90] = b
    s[90] = n_used
    return s


def unpack_system(s):
    s = np.asarray(s, dtype=np.float64)
    A = np.zeros((12, 12))
    A[np.triu_indices(12)] = s[:78]
    A = A + np.triu(A, 1).T
    return A, s[78:90].copy(), int(round(float(s[90])))


def allreduce_system(system_tensor, group=None):
    """The one collective of the path: in-place sum of the packed system across ranks."""
    import torch.distributed as dist
    dist.all_reduce(system_tensor, op=dist.ReduceOp.SUM, group=group)
    return system_tensor


class ShardedGnSolver:
    """The keypoint-sharded GN loop of one rank. Default: the whole loop runs inside libctgn (ctgn_solve_sharded) with ONE
    ncclAllReduce of the packed system per iteration issued from C on the handle's stream — no Python between the launches;
    torch.distributed is only used to hand rank 0's RCCL unique id to the other ranks. `library_collective=False` keeps the
    earlier variant (stepwise C-ABI calls with torch.distributed.all_reduce in between) for comparison."""

    def __init__(self, voxel_map, group=None, library_collective: bool = True):
        import torch
        import torch.distributed as dist
        from .registration import GnSolver
        self.torch = torch
        self.group = group
        self.library_collective = library_collective
        self.solver = GnSolver(voxel_map)
        self.system = None
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if library_collective:
            box = [GnSolver.dist_unique_id() if rank == 0 else None]
            if dist.is_initialized() and world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            self.solver.dist_init(rank, world, box[0])
        else:
            self.system = torch.zeros(SYSTEM_DOUBLES, dtype=torch.float64, device="cuda")
            self.solver.set_stream(torch.cuda.current_stream().cuda_stream)
            self.solver.gn_set_system_buffer(self.system.data_ptr())

    def close(self):
        """Release the communicator / give the library its own packed-system buffer back (the tensor may be freed afterwards)."""
        if self.solver is not None:
            if self.library_collective:
                self.solver.dist_shutdown()
            else:
                self.solver.gn_set_system_buffer(None)
            self.solver = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_keypoints(self, raw, world, t):
        self.solver.set_keypoints(raw, world, t)

    def set_keypoints_of_scan(self, raw, world, t):
        """The WHOLE scan on every rank: the library sorts it by home voxel and keeps this rank's contiguous chunk (ctgn_set_keypoints_sharded).
        Returns the chunk's indices into the scan."""
        import torch.distributed as dist
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        world_size = dist.get_world_size(self.group) if dist.is_initialized() else 1
        return self.solver.set_keypoints_sharded(raw, world, t, rank, world_size)

    def solve(self, pose14, t_begin_end, options, motion_model=None):
        s = self.solver
        if self.library_collective:
            return s.solve_sharded(pose14, t_begin_end, options, motion_model)
        # fail TOGETHER (as ctgn_solve_sharded does): a rank that cannot start, or that fails in the middle of the loop, still takes part
        # in every remaining exchange with a poisoned count (-1e300) — the peers' solve step sees the negative sum and stops with an
        # error too instead of waiting in the all-reduce for ever
        def join_poisoned(exchanges_left):
            for _ in range(exchanges_left):
                self.system.zero_()
                self.system[90] = -1e300
                allreduce_system(self.system, self.group)

        iters = int(options.num_iters_icp)
        try:
            s.gn_begin(pose14, t_begin_end, options, motion_model)
        except Exception:
            join_poisoned(iters)
            raise
        for it in range(iters):
            try:
                s.gn_accumulate()                  # local shard -> packed system in self.system
            except Exception:
                join_poisoned(iters - it)
                raise
            allreduce_system(self.system, self.group)
            try:
                s.gn_solve_update()                # identical on every rank
            except Exception:
                join_poisoned(iters - it - 1)
                raise
        return s.gn_end()
