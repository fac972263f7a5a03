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
    """Runs the stepwise C-ABI loop (ctgn_gn_begin / accumulate / solve_update / end) with the all-reduce in between.
    The library's kernels are enqueued on torch's current stream so that NCCL's stream dependencies order them."""

    def __init__(self, voxel_map, group=None):
        import torch
        from .registration import GnSolver
        self.torch = torch
        self.group = group
        self.solver = GnSolver(voxel_map)
        self.system = torch.zeros(SYSTEM_DOUBLES, dtype=torch.float64, device="cuda")
        self.solver.set_stream(torch.cuda.current_stream().cuda_stream)
        self.solver.gn_set_system_buffer(self.system.data_ptr())

    def close(self):
        """Give the library its own packed-system buffer back (the tensor may be freed afterwards)."""
        if self.solver is not None:
            self.solver.gn_set_system_buffer(None)
            self.solver = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_keypoints(self, raw, world, t):
        self.solver.set_keypoints(raw, world, t)

    def solve(self, pose14, t_begin_end, options, motion_model=None):
        s = self.solver
        s.gn_begin(pose14, t_begin_end, options, motion_model)
        for _ in range(options.num_iters_icp):
            s.gn_accumulate()                      # local shard -> packed system in self.system
            allreduce_system(self.system, self.group)
            s.gn_solve_update()                    # identical on every rank
        return s.gn_end()
