"""GpuVoxelMap — the host-side mirror of the reference's `ISlamMap` subset for the GN path
(include/ct_icp/map.h:14-83 over include/SlamCore/experimental/map.h:19-46), backed by libctgn's device map.

Method names and argument meaning follow ct_icp::MultipleResolutionVoxelMap (map.h:96-605):
InsertPointCloud, RemoveElementsFarFromLocation, ClearMap, NumPoints, MapAsPointCloud,
SearchParamsFromRadiusSearch, RadiusSearch / ComputeNeighborhoods (batched, on the GPU).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from . import _lib as L


@dataclass
class ResolutionParam:
    resolution: float = 0.5
    min_distance_between_points: float = 0.1
    max_num_points: int = 40


@dataclass
class GpuVoxelMapOptions:
    """MultipleResolutionVoxelMap::Options (map.h:115-133) + the device ordinal. map_type string for the YAML
    selector (src/ct_icp/map.cpp:68-77) would be "GPU_VOXEL_HASHMAP"."""
    resolutions: List[ResolutionParam] = field(default_factory=lambda: [ResolutionParam(0.2, 0.03, 50),
                                                                        ResolutionParam(0.5, 0.1, 40),
                                                                        ResolutionParam(1.5, 0.15, 40)])
    default_radius: float = 0.8
    max_frames_to_keep: int = 100
    device: int = 0                  # -1: host-only mirror (no queries possible); tests of the insert rule only
    device_updates: bool = False     # True: insert / evict rules run on the GPU (no host mirror), SURVEY 8f row 1
    initial_voxel_capacity: int = 0

    @staticmethod
    def Type() -> str:
        return "GPU_VOXEL_HASHMAP"


def _as_points(xyz):
    a = np.asarray(xyz)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    a = a.reshape(-1, 3) if a.ndim != 2 else a
    if a.strides[1] != a.itemsize:
        a = np.ascontiguousarray(a)
    return a


class GpuVoxelMap:
    def __init__(self, options: GpuVoxelMapOptions | None = None):
        self.options = options or GpuVoxelMapOptions()
        lib = L.lib()
        mo = L.MapOptions()
        lib.ctgn_map_options_default(C.byref(mo))
        mo.num_resolutions = len(self.options.resolutions)
        mo.device = self.options.device
        mo.default_radius = self.options.default_radius
        mo.initial_voxel_capacity = self.options.initial_voxel_capacity
        for i, r in enumerate(self.options.resolutions):
            mo.resolutions[i] = L.ResolutionParam(r.resolution, r.min_distance_between_points, r.max_num_points, 0)
        h = C.c_void_p()
        st = lib.ctgn_create(C.byref(mo), C.byref(h))
        if st != L.OK:
            raise L.CtgnError(st, lib.ctgn_status_string(st).decode())
        self._h = h
        if self.options.device_updates:
            L.check(self._h, lib.ctgn_map_set_update_mode(self._h, 1))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and L._lib is not None:
            L._lib.ctgn_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    # ---- update API -------------------------------------------------------------------------------------
    def InsertPointCloud(self, world_points) -> np.ndarray:
        """Insert world points in every resolution (map.h:153-254 / :261-293). Returns the mask of points that
        were kept by at least one resolution (the reference's out_selected_points)."""
        if L.is_device_tensor(world_points):        # device-resident points (torch CUDA tensor): nothing crosses PCIe
            import torch
            v = L.tensor_view(world_points)
            out_t = torch.zeros(len(world_points), dtype=torch.uint8, device=world_points.device)
            L.check(self._h, L.lib().ctgn_map_insert(self._h, v.base, v.stride_bytes, v.dtype, len(world_points),
                                                    C.cast(out_t.data_ptr(), C.POINTER(C.c_uint8))))
            return out_t.bool()
        a = _as_points(world_points)
        out = np.zeros(len(a), dtype=np.uint8)
        dt = L.CTGN_F64 if a.dtype == np.float64 else L.CTGN_F32
        L.check(self._h, L.lib().ctgn_map_insert(self._h, a.ctypes.data, a.strides[0], dt, len(a),
                                                out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.astype(bool)

    def RemoveElementsFarFromLocation(self, location, distance: float):
        loc = np.ascontiguousarray(location, dtype=np.float64)
        L.check(self._h, L.lib().ctgn_map_remove_far(self._h, loc.ctypes.data_as(C.POINTER(C.c_double)), float(distance)))

    def ClearMap(self):
        L.check(self._h, L.lib().ctgn_map_clear(self._h))

    def NumPoints(self) -> int:
        n = C.c_uint64()
        L.check(self._h, L.lib().ctgn_map_num_points(self._h, C.byref(n)))
        return n.value

    def NumVoxels(self, resolution_index: int = 0) -> int:
        n = C.c_uint64()
        L.check(self._h, L.lib().ctgn_map_num_voxels(self._h, resolution_index, C.byref(n)))
        return n.value

    def MapAsPointCloud(self, resolution_index: int = 0) -> np.ndarray:
        n = C.c_uint64()
        L.check(self._h, L.lib().ctgn_map_export(self._h, resolution_index, None, 0, C.byref(n)))
        out = np.zeros((n.value, 3))
        L.check(self._h, L.lib().ctgn_map_export(self._h, resolution_index, out.ctypes.data_as(C.POINTER(C.c_double)),
                                                n.value, C.byref(n)))
        return out

    # ---- query API --------------------------------------------------------------------------------------
    def SearchParamsFromRadiusSearch(self, radius: float | None = None):
        mid, nb, res = C.c_int32(), C.c_int32(), C.c_double()
        L.check(self._h, L.lib().ctgn_map_search_params(self._h, -1.0 if radius is None else radius, C.byref(mid),
                                                       C.byref(res), C.byref(nb)))
        return mid.value, res.value, nb.value

    def ComputeNeighborhoods(self, queries, max_num_neighbors: int, radius: float | None = None):
        """Batched RadiusSearch on the GPU. Returns a list of (n_i, 3) arrays, farthest neighbour first
        (map.h:508-513)."""
        q = np.ascontiguousarray(np.asarray(queries, dtype=np.float64).reshape(-1, 3))
        out = np.zeros((len(q), max_num_neighbors, 3))
        cnt = np.zeros(len(q), dtype=np.int32)
        L.check(self._h, L.lib().ctgn_map_radius_search(self._h, q.ctypes.data_as(C.POINTER(C.c_double)), len(q),
                                                       -1.0 if radius is None else float(radius), max_num_neighbors,
                                                       out.ctypes.data_as(C.POINTER(C.c_double)),
                                                       cnt.ctypes.data_as(C.POINTER(C.c_int32))))
        return [out[i, :cnt[i]].copy() for i in range(len(q))]

    def RadiusSearch(self, query, radius: float, max_num_neighbors: int) -> np.ndarray:
        return self.ComputeNeighborhoods(np.asarray(query, float).reshape(1, 3), max_num_neighbors, radius)[0]

    def ComputeNeighborhood(self, query, max_num_neighbors: int) -> np.ndarray:
        return self.ComputeNeighborhoods(np.asarray(query, float).reshape(1, 3), max_num_neighbors)[0]

    def Sync(self):
        L.check(self._h, L.lib().ctgn_map_sync(self._h))
