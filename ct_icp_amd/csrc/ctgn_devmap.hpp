// ctgn_devmap.hpp — device-resident maintenance of the voxel map:
// insert with per-voxel min-distance test + capacity cap (reference include/ct_icp/map.h:261-293) and far-voxel
// eviction (:305-322) executed ON the GPU, same slot / block layout as the host mirror (ctgn_map.hpp), so the query
// kernels do not care which side maintains the map. Opt-in per handle (ctgn_map_set_update_mode).
//
// Why the result is identical to the reference's sequential insertion: a point only interacts with the points of its own
// voxel, so voxels are independent; the batch is stably sorted by voxel key (ctgn_sort.hpp: hand-written LSD radix sort, values = original
// index) and ONE thread walks a voxel's run in original order applying the reference's rule against the points already
// in the block. Only block ids differ from a host-maintained map, and those are not observable.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ctgn_map.hpp"

namespace ctgn {

// scratch of the multi-block radix sort (kernels: ctgn_sort.hpp, included by ctgn_devmap.hip only)
struct SortScratch {
    uint32_t *hist = nullptr;                  // [256][columns] digit counts -> exclusive offsets
    unsigned long long *bits = nullptr;        // [2]: OR of all keys, AND of all keys
};

struct DevCounters {
    unsigned long long num_voxels, num_tombs, num_points;
    unsigned int next_block;     // bump allocator high-water mark
    int free_top;                // stack pointer of the free-block list
    unsigned int range_error;    // a point fell outside the 21-bit key range
    unsigned int overflow;       // table / block pool exhausted (host sized them: must stay 0)
};

struct DevLevel {
    // geometry (host copies)
    double resolution = 0.5, min_distance = 0.1;
    int blk = 40;
    // device storage
    Slot *slots = nullptr;
    uint64_t slots_cap = 0;              // power of two
    double *blocks = nullptr;
    uint32_t nblocks_cap = 0;
    uint32_t *free_list = nullptr;       // nblocks_cap entries
    DevCounters *counters = nullptr;     // device
    DevCounters host{};                  // last read-back
};

struct DevMapScratch {                   // per handle, shared by the levels
    double *pts = nullptr;               // [3][cap] staged points
    uint64_t *keys = nullptr, *keys_alt = nullptr;
    uint32_t *idx = nullptr, *idx_alt = nullptr;
    uint8_t *inserted = nullptr;
    uint32_t *sel_out = nullptr;         // grid sampling: selected indices
    int *sel_count = nullptr;
    SortScratch sort;                    // histogram matrix + key-bit words of the multi-block radix sort (ctgn_sort.hpp)
    size_t cap = 0;
    size_t stride = 0;                   // distance between the x / y / z planes of the staged batch (<= cap; set by the stager)
    double *h_pts = nullptr;             // pinned
    uint8_t *h_inserted = nullptr;       // pinned
    // grid sampling: open-addressing table voxel key -> smallest point index (2 x cap slots, power of two)
    unsigned long long *gs_keys = nullptr;
    uint32_t *gs_first = nullptr;
    size_t gs_cap = 0;
    int *h_count = nullptr;              // pinned read-back word
};

// AdaptiveGridSamplingOptions (reference include/ct_icp/algorithm/sampling.h:13-26), passed to the key kernel by value
constexpr int AS_MAX_BANDS = 16;
struct AdaptiveBands {
    int num_bands;
    int num_points_per_voxel;
    double distance[AS_MAX_BANDS];
    double voxel_size[AS_MAX_BANDS];
};

// All functions return hipSuccess or the failing HIP error; they enqueue on `stream` and synchronise where a read-back
// is needed.
hipError_t devmap_level_init(DevLevel &L, double resolution, double min_distance, int blk, hipStream_t stream);
void devmap_level_free(DevLevel &L);
hipError_t devmap_level_clear(DevLevel &L, hipStream_t stream);
hipError_t devmap_scratch_reserve(DevMapScratch &S, size_t n);
void devmap_scratch_free(DevMapScratch &S);
// points already staged in S.pts (SoA, stride S.cap), n of them; ORs into S.inserted
hipError_t devmap_level_insert(DevLevel &L, DevMapScratch &S, size_t n, hipStream_t stream);
hipError_t devmap_level_remove_far(DevLevel &L, const double loc[3], double distance, hipStream_t stream);
// enqueue-only variants of the frame pipeline (no read-back; the location / the gate live on the device), and the read-back they defer
hipError_t devmap_level_insert_enqueue(DevLevel &L, DevMapScratch &S, size_t n, const int *skip, hipStream_t stream);
hipError_t devmap_level_remove_far_enqueue(DevLevel &L, const double *loc_dev, double distance, hipStream_t stream, const int *failed_dev = nullptr);
hipError_t devmap_level_remove_far_value_enqueue(DevLevel &L, const double loc[3], double distance, hipStream_t stream);   // host location, no read-back
hipError_t devmap_level_read_counters(DevLevel &L, hipStream_t stream);
// sub_sample_frame / grid_sampling (reference src/ct_icp/ct_icp.cpp:65-101): index of the first point of every voxel of
// the staged points (voxel = static_cast<short>(p / voxel_size) per axis). Output in ascending index order = the order in
// which the reference's loop first meets the voxels (its robin_map iteration order is unspecified). out_idx_host (host or
// device memory) must hold n entries.
hipError_t devmap_grid_sampling(DevMapScratch &S, size_t n, double voxel_size, uint32_t *out_idx_host, size_t *out_count,
                                hipStream_t stream);
// The two samplers of a frame chained on the device, nothing read back (the frame pipeline, ctgn_frame_register): level 1 =
// sub_sample_frame(frame, frame_voxel) (odometry.cpp:352; <= 0: every point kept), level 2 = grid_sampling(sampled frame, keypoints,
// keypoint_voxel) (odometry.cpp:527; <= 0: keypoints = sampled frame). Coordinate a of point i = pts[i * es + a * stride]
// (planes: es 1; x y z t records: es 4, stride 1). flag1 / flag2: n bytes, sel1 / sel2: n indices (ascending), counts: 2 device
// ints {n sampled, n keypoints}. Six launches: clear, hash + flags per level, one two-way ordered compaction. Needs
// devmap_scratch_reserve(S, n).
hipError_t devmap_frame_sampling(DevMapScratch &S, const double *pts, size_t stride, size_t es, size_t n, double frame_voxel,
                                 double keypoint_voxel, uint8_t *flag1, uint8_t *flag2, uint32_t *sel1, uint32_t *sel2, int *counts,
                                 hipStream_t stream);
// AdaptiveSamplePointsInGrid (reference include/ct_icp/algorithm/sampling.h:55-110): range band -> voxel size, the first
// num_points_per_voxel indices of every (band, voxel); order band, voxel (z, y, x), index; at most max_num_points + 1
// indices when max_num_points > 0 (the reference stops on size() > max). The band list must be validated by the caller.
hipError_t devmap_adaptive_sampling(DevMapScratch &S, size_t n, const AdaptiveBands &bands, int max_num_points, uint32_t *out_idx,
                                    size_t *out_count, hipStream_t stream);
// Keypoint order for the neighbour-search kernel: indices sorted by the home voxel of their world point (12 / 12 / 8 bits of
// x / y / z, wrapped — aliasing only interleaves far-apart regions). A permutation only: which wave works on which keypoint —
// results do not depend on it. `order` stays valid until the next call.
struct OrderScratch {
    uint32_t *keys = nullptr, *keys_alt = nullptr, *idx = nullptr, *order = nullptr;
    SortScratch sort;
    size_t cap = 0;
};
hipError_t order_scratch_reserve(OrderScratch &S, size_t n);       // allocations only (kept out of the solve)
hipError_t order_by_home_voxel(OrderScratch &S, const double *wx, const double *wy, const double *wz, size_t n, double resolution,
                               hipStream_t stream);
void order_scratch_free(OrderScratch &S);
// test hooks (ctgn_test_sort_pairs / ctgn_test_compact): the primitives of ctgn_sort.hpp on host arrays
hipError_t devmap_test_sort(const uint64_t *keys_host, size_t n, int key_bits, int key_bytes, uint32_t *order_host, hipStream_t stream);
hipError_t devmap_test_compact(const uint8_t *flags_host, size_t n, uint32_t *out_host, size_t *count, hipStream_t stream);
hipError_t devmap_level_export(DevLevel &L, double *out_xyz, uint64_t cap_points, uint64_t *out_n, hipStream_t stream);

}  // namespace ctgn
