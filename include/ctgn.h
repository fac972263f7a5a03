/*
 * ctgn.h — C ABI of libctgn.so: the MI355X (gfx950) Gauss–Newton CT-ICP registration path.
 *
 * This is the drop-in boundary for ONE hot path of jedeschaud/ct_icp:
 *   ct_icp::CT_ICP_Registration::DoRegisterGaussNewton   (reference src/ct_icp/ct_icp.cpp:709-996)
 * together with the read side of the local voxel map it queries
 *   ct_icp::MultipleResolutionVoxelMap::RadiusSearchInPlace (reference include/ct_icp/map.h:449-514)
 * and the write side needed to keep that map resident on the GPU
 *   InsertPointInVoxelMap / RemoveElementsFarFromLocation  (reference include/ct_icp/map.h:261-293, 305-322),
 * plus the rows either side of the path: grid sampling, full-scan undistortion, device-resident map
 * maintenance, and the robust-loss route the shipped configs select
 *   ct_icp::CT_ICP_Registration::DoRegisterCeres          (reference src/ct_icp/ct_icp.cpp:457-707; ctgn_register_robust),
 * those rows chained with the scan resident on the device — the data-parallel side of
 *   ct_icp::Odometry::DoRegister                          (reference src/ct_icp/odometry.cpp:333-501, 936-952; ctgn_frame_*),
 * and the keypoint-sharded multi-GPU loop with its one all-reduce issued by the library (ctgn_dist_*, ctgn_solve_sharded).
 *
 * The reference has no C ABI / FFI for this path (it is C++ member calls inside libCT_ICP.so), so the
 * entry points below are what a maintainer would bind from
 *   - a `GpuVoxelMap : ct_icp::ISlamMap` (reference include/ct_icp/map.h:14-83) -> ctgn_map_* calls
 *   - the `case GN:` / `case CERES:` arms of SELECT_SOLVER (reference src/ct_icp/ct_icp.cpp:1003-1014) -> ctgn_register / ctgn_register_robust
 *   - Odometry::DoRegister's sampling / registration / undistortion / map update -> ctgn_frame_register + ctgn_frame_update_map
 * INTEGRATION.md shows that binding; ct_icp_amd/cpp/ct_icp_gpu.hpp is the C++ adapter.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; every function returns a ctgn_status
 *     (0 = ok, <0 = error) and never throws or aborts across the boundary.
 *   - poses are 7 doubles (qx, qy, qz, qw, tx, ty, tz): Eigen's coeffs() order, the same order as
 *     slam::TSE3::operator[] (reference include/SlamCore/types.h:378-385).
 *   - all host buffers are borrowed for the duration of the call only.
 *   - point / timestamp views and point outputs may address DEVICE memory of the handle's GPU instead (detected with
 *     hipPointerGetAttributes): ctgn_set_keypoints, ctgn_get_world_points, ctgn_register*, ctgn_transform_points,
 *     ctgn_grid_sampling (input and out_indices) and, with the device-resident map, ctgn_map_insert (input and
 *     out_inserted). Nothing is then staged through the host; all views of one call must live on the same side.
 *   - a handle owns one HIP device + one HIP stream and is NOT re-entrant (the reference's Odometry is
 *     not thread-safe either, include/ct_icp/odometry.h:275-287).
 *   - there is NO CPU fallback: every call fails with CTGN_ERR_NO_DEVICE when no gfx950 device exists.
 */
#ifndef CTGN_H
#define CTGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history: 6 (round 6) = ctgn_frame_outputs grew keypoint_world_* at its end + ctgn_frame_begin / _try_register / _undistort;
 * 5 (round 5) = ctgn_count_traffic, ctgn_kernel_timing_split and ctgn_set_variant left this header for ctgn_internal.h (the symbols are
 * still exported; they are measurement hooks, not part of the contract). */
#define CTGN_ABI_VERSION 6
#define CTGN_MAX_RESOLUTIONS 8
/* The hard-coded "not enough keypoints" bound of the reference (src/ct_icp/ct_icp.cpp:860). */
#define CTGN_MIN_KEYPOINTS_USED 100
/* Packed normal equations: 78 upper-triangular JtJ + 12 Jtr + 1 count, padded to 96 doubles. */
#define CTGN_SYSTEM_DOUBLES 96
/* Upper bound of CTICPOptions::max_number_neighbors supported by the kernels. */
#define CTGN_MAX_NEIGHBORS 32

typedef struct ctgn_context *ctgn_handle;

typedef enum {
    CTGN_OK = 0,
    CTGN_ERR_INVALID_ARGUMENT = -1,
    CTGN_ERR_NO_DEVICE = -2,       /* no HIP device / not gfx950: there is no CPU fallback           */
    CTGN_ERR_HIP = -3,             /* a HIP runtime call failed; see ctgn_last_error()                */
    CTGN_ERR_OUT_OF_MEMORY = -4,
    CTGN_ERR_TIMESTAMP_RANGE = -5, /* a keypoint timestamp lies outside [t_begin, t_end]: the reference
                                      glog-CHECK-aborts here (include/SlamCore/types.h:456)           */
    CTGN_ERR_VOXEL_RANGE = -6,     /* voxel coordinate outside the int16 sweep range of the reference
                                      (`short kxx`, include/ct_icp/map.h:470-472)                     */
    CTGN_ERR_UNSUPPORTED = -7,
    CTGN_ERR_SOLVER = -8           /* robust route: the inner solver reports an unusable solution; the reference
                                      throws std::runtime_error("Error During Optimization"), ct_icp.cpp:628-631 */
} ctgn_status;

typedef enum { CTGN_F32 = 0, CTGN_F64 = 1 } ctgn_dtype;

/* -------------------------------------------------------------------------------------------------
 * Map: mirrors ct_icp::MultipleResolutionVoxelMap::Options / ResolutionParam
 * (reference include/ct_icp/map.h:109-125).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    double resolution;                  /* voxel edge (m)                                   */
    double min_distance_between_points; /* insert only if farther than this from all others */
    int32_t max_num_points;             /* per-voxel capacity                               */
    int32_t _pad;
} ctgn_resolution_param;

typedef struct {
    int32_t num_resolutions;
    int32_t device;                     /* HIP device ordinal                                          */
    double default_radius;              /* radius used by ComputeNeighborhood (map.h:527-530)         */
    ctgn_resolution_param resolutions[CTGN_MAX_RESOLUTIONS]; /* ascending resolution, as the reference */
    uint64_t initial_voxel_capacity;    /* 0 = default; tables grow on demand                          */
} ctgn_map_options;

/* Fills the reference's defaults: {0.2,0.03,50},{0.5,0.1,40},{1.5,0.15,40}, radius 0.8 (map.h:117-125). */
void ctgn_map_options_default(ctgn_map_options *opts);

/* Create a context (device map + solver workspace). Replaces MakeMapFromOptions (map.h:127-133). */
ctgn_status ctgn_create(const ctgn_map_options *opts, ctgn_handle *out);
void ctgn_destroy(ctgn_handle h);
const char *ctgn_last_error(ctgn_handle h);          /* valid until the next call on h */
const char *ctgn_status_string(ctgn_status s);
int32_t ctgn_abi_version(void);

/* InsertPointCloud on world points (map.h:153-254 -> InsertPointInVoxelMap :261-293), applied to every
 * resolution. Points are read from a strided view (base + i*stride_bytes -> 3 x dtype).
 * out_inserted (optional, n bytes): 1 if the point was inserted in at least one resolution
 * (the reference's `selected_indices`, map.h:196-206). A point whose voxel coordinate does not fit the 21-bit key range, or that is
 * not finite, is SKIPPED (out_inserted = 0) and the call still returns CTGN_OK with a note in ctgn_last_error: the rest of the batch
 * is in the map, and an error return would invite a retry that inserts it twice (the reference has no such failure at all). */
ctgn_status ctgn_map_insert(ctgn_handle h, const void *xyz_base, size_t stride_bytes, ctgn_dtype dtype,
                            size_t n, uint8_t *out_inserted);
/* RemoveElementsFarFromLocation (map.h:305-322): drops a voxel iff its FIRST point is farther than
 * `distance` from `location`. */
ctgn_status ctgn_map_remove_far(ctgn_handle h, const double location[3], double distance);
/* ClearMap (map.h:296). */
ctgn_status ctgn_map_clear(ctgn_handle h);
/* NumPoints (map.h:341-347): summed over all resolutions, as the reference does. */
ctgn_status ctgn_map_num_points(ctgn_handle h, uint64_t *out);
ctgn_status ctgn_map_num_voxels(ctgn_handle h, int32_t resolution_index, uint64_t *out);
/* SearchParamsFromRadiusSearch (map.h:416-432): which resolution a radius selects and the sweep half-width. */
ctgn_status ctgn_map_search_params(ctgn_handle h, double radius, int32_t *map_id, double *voxel_resolution,
                                   int32_t *voxel_neighborhood);
/* MapAsPointCloud for one resolution (map.h:349-407, xyz only). Call with out=NULL to get the count. */
ctgn_status ctgn_map_export(ctgn_handle h, int32_t resolution_index, double *out_xyz, uint64_t capacity_points,
                            uint64_t *out_num_points);
/* Who maintains the map. 0 (default): the host mirror applies the insert / evict rules and
 * the device copy is updated by deltas. 1: the rules run ON the GPU (stable radix sort of the batch by voxel key, one
 * thread per voxel walks its points in original order) — same results, no host mirror. Only on an empty map. */
ctgn_status ctgn_map_set_update_mode(ctgn_handle h, int32_t device_updates);
/* Push pending host-side map edits to the device now (otherwise done lazily by the next query). */
ctgn_status ctgn_map_sync(ctgn_handle h);

/* ComputeNeighborhoods (map.h:532-541) / RadiusSearch (map.h:516-522) for a batch of queries, ON THE GPU.
 * Output layout per query i: out_count[i] neighbours, out_xyz[(i*max_num_neighbors + j)*3 ..] for
 * j < out_count[i], FARTHEST FIRST (the reference drains a max-heap from top(), map.h:508-513).
 * radius <= 0 selects options.default_radius. */
ctgn_status ctgn_map_radius_search(ctgn_handle h, const double *queries_xyz, size_t n, double radius,
                                   int32_t max_num_neighbors, double *out_xyz, int32_t *out_count);

/* -------------------------------------------------------------------------------------------------
 * Solver: the GN-relevant fields of ct_icp::CTICPOptions (reference include/ct_icp/ct_icp.h:56-153;
 * the fields DoRegisterGaussNewton actually reads, src/ct_icp/ct_icp.cpp:727,743,762,803,866,978).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t num_iters_icp;               /* default 5                                   */
    int32_t min_number_neighbors;        /* default 20                                  */
    int32_t max_number_neighbors;        /* default 20, <= CTGN_MAX_NEIGHBORS           */
    int32_t debug_print;                 /* print the soft-failure message to stdout    */
    double max_dist_to_plane_ct_icp;     /* default 0.3                                 */
    double threshold_orientation_norm;   /* default 1e-4 (stop on ||x||_2 of all 12)    */
} ctgn_options;

void ctgn_options_default(ctgn_options *opts);

/* PreviousFrameMotionModel terms (reference src/ct_icp/ct_icp.cpp:885-910,
 * include/ct_icp/motion_model.h:42-58,72). Pass NULL when the motion model is null or not a
 * PreviousFrameMotionModel. */
typedef struct {
    double beta_location_consistency;    /* ALPHA_C */
    double beta_constant_velocity;       /* ALPHA_E */
    double previous_begin_tr[3];         /* PreviousFrame().BeginTr() */
    double previous_end_tr[3];           /* PreviousFrame().EndTr()   */
} ctgn_motion_prior;

/* ICPSummary (reference include/ct_icp/ct_icp.h:155-169) + device timings. */
typedef struct {
    int32_t success;                     /* ICPSummary::success                                  */
    int32_t num_residuals_used;          /* keypoints that passed the gates in the last iteration */
    int32_t num_iters;                   /* GN iterations executed (reference leaves this 0)      */
    int32_t _pad;
    double duration_total_ms;            /* host wall time of the call                            */
    double duration_device_ms;           /* HIP-event time of the iteration loop                  */
    double last_step_norm;               /* ||x||_2 of the last solve                             */
    /* ICPSummary::duration_init / avg_duration_neighborhood / avg_duration_solve / avg_duration_iter (ct_icp.h:164-168; the
     * reference fills them on its CERES route, :670-672,688-691). Per executed iteration, from the device's constant 100 MHz
     * clock stamped by the kernels themselves (no extra launches or events): neighbourhood = start of the neighbour-search
     * kernel -> start of the solve kernel (search + normals + residuals + reduction), solve = the solve kernel, iter = both.
     * The robust route reports the totals of its ICP iterations the same way through duration_device_ms only. */
    double duration_init_ms;
    double avg_duration_neighborhood_ms;
    double avg_duration_solve_ms;
    double avg_duration_iter_ms;
    char error_log[256];                 /* ICPSummary::error_log (same text as ct_icp.cpp:862-863) */
} ctgn_summary;

/* A strided read-only view of N elements, the C image of slam::ProxyView
 * (reference include/SlamCore/data/view.h:98-116). */
typedef struct {
    const void *base;
    size_t stride_bytes;
    ctgn_dtype dtype;
    int32_t _pad;
} ctgn_view;

/* Upload a keypoint set (raw xyz, world xyz, timestamp) to the device. */
ctgn_status ctgn_set_keypoints(ctgn_handle h, ctgn_view raw_xyz, ctgn_view world_xyz, ctgn_view timestamps,
                               size_t n);
/* Keep a device-side copy of the world points of every later ctgn_set_keypoints so that ctgn_rewind_keypoints can put them back
 * without another upload: a registration retried on the SAME keypoints from their initial world points — what Odometry::TryRegister
 * does when it re-registers a frame with more robust settings (the do / while around TryRegister, reference src/ct_icp/odometry.cpp:794-845) — and what bench.py does
 * to time fresh solves with the inputs resident in HBM. Off by default (it costs one device-to-device copy per upload). */
ctgn_status ctgn_set_rewind(ctgn_handle h, int32_t enable);
/* Restore the resident keypoints' world points to what the last ctgn_set_keypoints uploaded (enqueued on the handle's stream, no
 * synchronisation) and forget everything the previous solve left per keypoint (carried-over search bounds, ordering). Needs
 * ctgn_set_rewind(h, 1) before that upload. */
ctgn_status ctgn_rewind_keypoints(ctgn_handle h);
/* Run the GN loop on the resident keypoints. pose_io = begin(7) | end(7); t_begin_end = dest_timestamp
 * of begin_pose / end_pose. Replaces DoRegisterGaussNewton (ct_icp.cpp:709-996). */
ctgn_status ctgn_solve(ctgn_handle h, double pose_io[14], const double t_begin_end[2],
                       const ctgn_options *opts, const ctgn_motion_prior *prior, ctgn_summary *summary);
/* Copy the (re-transformed) world points back into a strided host view (ct_icp.cpp:964-966). */
ctgn_status ctgn_get_world_points(ctgn_handle h, void *world_base, size_t stride_bytes, ctgn_dtype dtype,
                                  size_t n);

/* Continuous-time transform of an arbitrary point set with a begin|end pose — the step right after the path:
 * the full-scan undistortion of Odometry::DoRegister (reference src/ct_icp/odometry.cpp:461-486,
 * out[i] = begin.InterpolatePose(end, t[i]) * raw[i], include/SlamCore/types.h:413-419,453-470). Timestamps outside
 * [t_begin, t_end] return CTGN_ERR_TIMESTAMP_RANGE (the reference CHECK-aborts). In and out may alias. */
ctgn_status ctgn_transform_points(ctgn_handle h, ctgn_view raw_xyz, ctgn_view timestamps, size_t n, const double pose[14],
                                  const double t_begin_end[2], void *out_base, size_t out_stride_bytes, ctgn_dtype out_dtype);

/* sub_sample_frame / grid_sampling on the GPU (reference src/ct_icp/ct_icp.cpp:65-101): keeps the
 * FIRST point of every voxel of size `voxel_size`, voxel = static_cast<short>(p / voxel_size) per axis. Writes the kept
 * indices (at most n) to out_indices and their number to out_count. Order: ascending index, i.e. the order in which the
 * reference's loop first meets the voxels (its output order is the unspecified tsl::robin_map iteration order). */
ctgn_status ctgn_grid_sampling(ctgn_handle h, ctgn_view xyz, size_t n, double voxel_size, uint32_t *out_indices, size_t *out_count);

/* AdaptiveGridSamplingOptions (reference include/ct_icp/algorithm/sampling.h:13-26): distance / voxel_size are the pairs of
 * distance_voxel_size, ascending in distance; the voxel size of the last pair is never used. */
#define CTGN_ADAPTIVE_MAX_BANDS 16
typedef struct ctgn_adaptive_sampling_options {
    int32_t num_points_per_voxel;   /* 1 */
    int32_t max_num_points;         /* -1: unlimited */
    int32_t num_bands;              /* 6 */
    int32_t reserved;
    double distance[CTGN_ADAPTIVE_MAX_BANDS];     /* 0.5, 2, 4, 8, 16, 200 */
    double voxel_size[CTGN_ADAPTIVE_MAX_BANDS];   /* 0.1, 0.2, 0.4, 0.8, 1.6, -1 */
} ctgn_adaptive_sampling_options;
void ctgn_adaptive_sampling_options_default(ctgn_adaptive_sampling_options *o);

/* AdaptiveSamplePointsInGrid on the GPU (reference include/ct_icp/algorithm/sampling.h:55-110 — the keypoint sampling of
 * `sampling: ADAPTIVE`, src/ct_icp/odometry.cpp:539-545): a point at range d = |p| with distance[0] <= d < distance[last] falls
 * in band j = (first pair with distance >= d) - 1 and in voxel int(p / voxel_size[j]) per axis; every (band, voxel) keeps its
 * first num_points_per_voxel indices. Output order: band, then voxel (z, y, x ascending), then index (the reference walks
 * std::unordered_maps, order unspecified); with max_num_points > 0 at most max_num_points + 1 indices are written — the
 * reference stops on `size() > max` (:96-106). d == distance[0] exactly is undefined behaviour in the reference (entry -1)
 * and is dropped here. out_indices (host or device memory) must hold n entries. CTGN_ERR_INVALID_ARGUMENT for a band list that
 * is not ascending, has fewer than 2 or more than CTGN_ADAPTIVE_MAX_BANDS pairs, a used voxel size <= 0, or a band whose
 * voxel coordinates would not fit 20 bits (distance[j + 1] / voxel_size[j] >= 2^19). */
ctgn_status ctgn_adaptive_sampling(ctgn_handle h, ctgn_view xyz, size_t n, const ctgn_adaptive_sampling_options *opts,
                                   uint32_t *out_indices, size_t *out_count);

/* One-shot drop-in for `case GN:` of SELECT_SOLVER (ct_icp.cpp:1008-1014) =
 * ctgn_set_keypoints + ctgn_solve + ctgn_get_world_points (world points are rewritten in place). */
ctgn_status ctgn_register(ctgn_handle h, ctgn_view raw_xyz, void *world_base, size_t world_stride_bytes,
                          ctgn_dtype world_dtype, ctgn_view timestamps, size_t n, double pose_io[14],
                          const double t_begin_end[2], const ctgn_options *opts,
                          const ctgn_motion_prior *prior, ctgn_summary *summary);

/* -------------------------------------------------------------------------------------------------
 * Robust-loss (CERES-profile) route — SURVEY.md section 8f row 4.
 * Replaces CT_ICP_Registration::DoRegisterCeres (reference src/ct_icp/ct_icp.cpp:457-707) for
 * parametrization = CONTINUOUS_TIME and distance = POINT_TO_PLANE, the configuration of every shipped config
 * (`solver: CERES`, config/odometry/driving_config.yaml:52-89): same neighbour search kernel as the GN route, then
 * planarity / neighbourhood weights (:525-532,574-579), the robust loss (:170-187), the max_num_residuals cap
 * (:415-426), the PreviousFrameMotionModel regularisers (src/ct_icp/motion_model.cpp:12-61) and an on-device
 * restatement of Ceres' Levenberg-Marquardt trust-region minimiser (ls_max_num_iters iterations per ICP iteration).
 * Equivalence with a Ceres build is algorithmic, not bit-exact: Ceres is a third-party dependency absent from the
 * reference tree.
 * ---------------------------------------------------------------------------------------------- */
typedef enum {                           /* ct_icp::LEAST_SQUARES, include/ct_icp/ct_icp.h:41-47 */
    CTGN_LOSS_STANDARD = 0, CTGN_LOSS_CAUCHY = 1, CTGN_LOSS_HUBER = 2, CTGN_LOSS_TOLERANT = 3, CTGN_LOSS_TRUNCATED = 4
} ctgn_loss;

typedef struct {                         /* CTICPOptions fields read by DoRegisterCeres (ct_icp.h:58-132) */
    int32_t num_iters_icp;               /* default 5                                                     */
    int32_t min_number_neighbors;        /* default 20; also the soft-failure bound on the residual count  */
    int32_t max_number_neighbors;        /* default 20 (INeighborStrategyOptions::max_num_neighbors)       */
    int32_t debug_print;
    int32_t max_num_residuals;           /* default -1 = no cap                                            */
    int32_t loss_function;               /* ctgn_loss, default CAUCHY                                      */
    int32_t ls_max_num_iters;            /* default 1                                                      */
    int32_t num_closest_neighbors;       /* default 1, <= min_number_neighbors                             */
    double weight_alpha;                 /* default 0.9                                                    */
    double weight_neighborhood;          /* default 0.1                                                    */
    double power_planarity;              /* default 2.0                                                    */
    double max_dist_to_plane_ct_icp;     /* default 0.3 (scale of the neighbourhood weight only)           */
    double ls_sigma;                     /* default 0.1                                                    */
    double ls_tolerant_min_threshold;    /* default 0.05                                                   */
    double threshold_orientation_norm;   /* default 1e-4, DEGREES on this route (ct_icp.cpp:645-646,662)   */
    double threshold_translation_norm;   /* default 1e-3                                                   */
} ctgn_robust_options;

void ctgn_robust_options_default(ctgn_robust_options *opts);

typedef struct {                         /* PreviousFrameMotionModel::Options + previous frame (motion_model.h:42-58) */
    double beta_location_consistency;    /* default 0.001 */
    double beta_constant_velocity;       /* default 0.001 */
    double beta_small_velocity;          /* default 0     */
    double beta_orientation_consistency; /* default 0     */
    double previous_begin_tr[3];         /* PreviousFrame().BeginTr()  */
    double previous_end_tr[3];           /* PreviousFrame().EndTr()    */
    double previous_end_quat[4];         /* PreviousFrame().EndQuat()  */
} ctgn_robust_prior;

/* Run the robust ICP loop on the resident keypoints (ctgn_set_keypoints; their world coordinates are ignored and
 * overwritten: the reference re-derives them from the poses at the top of every iteration, ct_icp.cpp:499-515,537).
 * summary->num_residuals_used = residual blocks of the last iteration; a residual count below
 * min_number_neighbors is the reference's soft failure (success = 0, same error_log text as :612-618). */
ctgn_status ctgn_solve_robust(ctgn_handle h, double pose_io[14], const double t_begin_end[2],
                              const ctgn_robust_options *opts, const ctgn_robust_prior *prior, ctgn_summary *summary);
/* One-shot drop-in for `case CERES:` of SELECT_SOLVER (ct_icp.cpp:1003-1007). */
ctgn_status ctgn_register_robust(ctgn_handle h, ctgn_view raw_xyz, void *world_base, size_t world_stride_bytes,
                                 ctgn_dtype world_dtype, ctgn_view timestamps, size_t n, double pose_io[14],
                                 const double t_begin_end[2], const ctgn_robust_options *opts,
                                 const ctgn_robust_prior *prior, ctgn_summary *summary);

typedef struct {                         /* state of the last ctgn_solve_robust (tests / diagnostics) */
    double cost;                         /* 1/2 sum rho(r^2) + regularisers at the last evaluated pose          */
    double radius;                       /* trust-region radius of the last inner solve                        */
    double diff_rot_deg, diff_trans;     /* pose change of the last ICP iteration (the stop test's inputs)     */
    int32_t num_residuals;               /* residual blocks of the last ICP iteration                          */
    int32_t ls_iterations;               /* Levenberg-Marquardt iterations over all ICP iterations             */
    int32_t ls_accepted;                 /* of which accepted                                                  */
    int32_t converged;                   /* the ICP stop test fired                                            */
    double JtJ[144];                     /* loss-corrected normal equations of the last evaluation, row-major, */
    double Jtr[12];                      /* tangent order: begin_quat | end_quat | begin_t | end_t             */
    uint64_t step_cycles[8];             /* shader clocks of the last LM step kernel, per stage (diagnostics)  */
} ctgn_robust_report;
ctgn_status ctgn_robust_get_report(ctgn_handle h, ctgn_robust_report *out);
/* Residual blocks of the last ICP iteration, per keypoint (any pointer may be NULL): normal[3n], weight[n], alpha[n],
 * reference[3n] (the first of the num_closest_neighbors points), rank[n] (index of the keypoint's first block in the
 * problem, -1 = no block). */
ctgn_status ctgn_robust_get_blocks(ctgn_handle h, double *normal, double *weight, double *alpha, double *reference,
                                   int32_t *rank, size_t n);

/* -------------------------------------------------------------------------------------------------
 * Stepwise GN (for the keypoint-sharded multi-GPU mode: one all-reduce of the packed system between
 * accumulate and solve; the library itself never calls a collective).
 * ---------------------------------------------------------------------------------------------- */
ctgn_status ctgn_gn_begin(ctgn_handle h, const double pose[14], const double t_begin_end[2],
                          const ctgn_options *opts, const ctgn_motion_prior *prior);
/* Enqueue neighbour search + residual/Jacobian + reduction for the current pose; the packed system
 * (CTGN_SYSTEM_DOUBLES doubles: 78 JtJ upper | 12 Jtr | count | pad) is left in device memory. */
ctgn_status ctgn_gn_accumulate(ctgn_handle h);
/* Enqueue `iterations` whole GN iterations behind ctgn_gn_begin without synchronising: the fused launch sequence of ctgn_solve, or
 * (sharded != 0, after ctgn_dist_init) the one of ctgn_solve_sharded with its ncclAllReduce. ctgn_solve == begin + iterate(num_iters_icp)
 * + end. Lets a caller (bench.py) bracket exactly K iterations of an already running loop with its own timers. */
ctgn_status ctgn_gn_iterate(ctgn_handle h, int32_t iterations, int32_t sharded);
/* Device address of the packed system (valid for the life of the handle, or until ctgn_gn_set_system_buffer). */
ctgn_status ctgn_gn_system_device_ptr(ctgn_handle h, void **out_device_ptr);
/* Make the library keep the packed system in a caller-owned device buffer of CTGN_SYSTEM_DOUBLES doubles (e.g. a
 * torch tensor that torch.distributed all-reduces in place). NULL restores the handle's own buffer. */
ctgn_status ctgn_gn_set_system_buffer(ctgn_handle h, void *device_ptr);
/* Enqueue normalise + motion prior + 12x12 solve + pose update + stop test (ct_icp.cpp:877-980). */
ctgn_status ctgn_gn_solve_update(ctgn_handle h);
/* Synchronise, re-transform the world points with the final pose, return pose + summary. */
ctgn_status ctgn_gn_end(ctgn_handle h, double pose_out[14], ctgn_summary *summary);

/* The keypoint-sharded loop with the collective issued by the library (no host code between the launches of an iteration):
 * map replicated on every rank, each rank uploads ITS contiguous shard of the keypoints (ctgn_set_keypoints), then per iteration
 * accumulate -> ncclAllReduce(sum, CTGN_SYSTEM_DOUBLES doubles, in place, on the handle's stream) -> solve / update on every rank
 * (identical reduced input => identical poses, no broadcast). RCCL is bound at run time (dlopen "librccl.so.1"; shared with
 * torch.distributed's instance when that is loaded). Bootstrap: rank 0 calls ctgn_dist_unique_id and ships the 128 bytes to the
 * other ranks by whatever means the host has (the reference has no launcher to mirror: it is a single process), every rank
 * then calls ctgn_dist_init (collective). Replaces nothing in the reference: ct_icp.cpp:843-850 sums over ALL keypoints and
 * :877-882 needs the global count — this is that sum across GPUs. */
#define CTGN_DIST_ID_BYTES 128
ctgn_status ctgn_dist_unique_id(uint8_t out[CTGN_DIST_ID_BYTES]);
ctgn_status ctgn_dist_init(ctgn_handle h, int32_t rank, int32_t world_size, const uint8_t id[CTGN_DIST_ID_BYTES]);
ctgn_status ctgn_dist_shutdown(ctgn_handle h);
/* Config D's partition done by the library ("keypoints split into G contiguous chunks after the voxel-key sort"): every
 * rank passes the WHOLE scan (n keypoints; host or device views as for ctgn_set_keypoints); the library sorts it on the device by the home
 * voxel of the world points at the searched resolution (stable, deterministic: every rank computes the same order) and keeps the rank-th of
 * world_size contiguous, balanced chunks as this handle's resident keypoints, already in the order the kernels want (no second sort).
 * shard_indices (host, nullable, >= n / world_size + 1 entries): index in the caller's arrays of resident keypoint j — what
 * ctgn_get_world_points's j-th row belongs to; *shard_n: how many this rank holds. The timestamp check of the solve uses the whole scan's
 * range, so every rank accepts or rejects the same scan (no rank is left waiting in the all-reduce). */
ctgn_status ctgn_set_keypoints_sharded(ctgn_handle h, ctgn_view raw_xyz, ctgn_view world_xyz, ctgn_view timestamps, size_t n,
                                       int32_t rank, int32_t world_size, uint32_t *shard_indices, size_t *shard_n);
ctgn_status ctgn_solve_sharded(ctgn_handle h, double pose_io[14], const double t_begin_end[2], const ctgn_options *opts,
                               const ctgn_motion_prior *prior, ctgn_summary *summary);
/* Non-blocking query of the device-side stop flag of the running GN loop (synchronises the stream). */
ctgn_status ctgn_gn_done(ctgn_handle h, int32_t *done);

/* Use an externally owned HIP stream (e.g. torch's current stream) instead of the handle's own. */
ctgn_status ctgn_set_stream(ctgn_handle h, void *hip_stream);
ctgn_status ctgn_get_stream(ctgn_handle h, void **hip_stream);

/* -------------------------------------------------------------------------------------------------
 * One frame of the odometry loop with the scan resident on the device: what Odometry::DoRegister does
 * around the path — InitializeFrame's sub_sample_frame + initial transform (reference src/ct_icp/odometry.cpp:333-382), TryRegister's
 * grid_sampling + Register + frame transform (:526-590), the undistortion of every point and of the sampled frame (:461-486) and
 * UpdateMap's RemoveElementsFarFromLocation + InsertPointCloud (:936-952) — chained on the handle's stream: the scan is uploaded
 * once, the two samplers, the keypoint gather, the GN (or robust) loop and both undistortions run without the host, ONE small
 * read-back (the two sampled counts) sizes the launches, and one copy returns the pose, the summary and the outputs that were asked
 * for. ctgn_frame_update_map then evicts and inserts from the device-resident undistorted frame: the reference decides in between
 * (AssessRegistration / insertion policy, host logic), which is why the frame is two calls; ctgn_frame is both with the
 * always_insert policy.
 *
 * The reference shuffles the frame before each sampler (its own std::mt19937, odometry.cpp:349,365). The first shuffle decides which
 * point of a voxel survives, so the caller passes it: `order` (host, n entries, a permutation of 0..n-1, or NULL = scan order) — the
 * frame is processed as points order[0], order[1], ... The second shuffle only randomises the reference's robin_map iteration order;
 * here the sampled frame is in ascending processing order, which is already a random order of space when `order` is a shuffle.
 * Every index this API reports is the caller's own point number.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    double frame_voxel_size;             /* sub_sample_frame's voxel (options.voxel_size / init_voxel_size); <= 0: keep all points */
    double sample_voxel_size;            /* grid_sampling's voxel for the keypoints; <= 0: keypoints = sampled frame (sampling NONE) */
    int32_t max_num_keypoints;           /* > 0: keep the first max_num_keypoints keypoints (odometry.cpp:549-552)            */
    int32_t override_timestamps;         /* 1: every point takes override_timestamp (registered_fid <= 1, odometry.cpp:357-361)  */
    double override_timestamp;
    uint64_t shuffle_seed;               /* != 0 and order == NULL: the first shuffle is made ON THE DEVICE — a keyed bijection of
                                            0..n-1 (six Feistel rounds with cycle walking, one thread per point, no host work): a uniformly
                                            random-looking processing order, reproducible from the seed, NOT the permutation libstdc++'s
                                            std::shuffle would make of the caller's engine (pass that one as `order` when the sampled set
                                            must be the reference's own). 0: scan order.                                          */
} ctgn_frame_options;
void ctgn_frame_options_default(ctgn_frame_options *o);

typedef struct {                         /* any pointer may be NULL = not wanted; host memory                                     */
    void *all_world_base;                /* n records: every scan point undistorted with the final poses (all_corrected_points)  */
    size_t all_world_stride_bytes;
    ctgn_dtype all_world_dtype;
    int32_t _pad0;
    uint32_t *sampled_indices;           /* capacity n: scan index of every point of the sampled frame                           */
    void *sampled_world_base;            /* capacity n records: the sampled frame undistorted (corrected_points)                 */
    size_t sampled_world_stride_bytes;
    ctgn_dtype sampled_world_dtype;
    int32_t _pad1;
    uint32_t *keypoint_indices;          /* capacity n: scan index of every keypoint                                             */
    uint64_t num_sampled;                /* out                                                                                  */
    uint64_t num_keypoints;              /* out                                                                                  */
    void *keypoint_world_base;           /* ctgn_frame_try_register only; capacity num_keypoints records: the keypoints' world points
                                            under the optimised poses (registration_summary.keypoints, odometry.cpp:597)         */
    size_t keypoint_world_stride_bytes;
    ctgn_dtype keypoint_world_dtype;
    int32_t _pad2;
    uint64_t num_keypoint_candidates;    /* out: keypoints before the max_num_keypoints cut (>= num_keypoints)                    */
} ctgn_frame_outputs;

/* Sampling -> keypoints -> registration -> undistortion of one scan. `robust` NULL: the GN route with `opts` (+ `prior`, may be
 * NULL); `robust` non-NULL: the robust-loss route (`opts` / `prior` ignored, `robust_prior` may be NULL). pose_io in: the initial
 * begin|end estimate (trajectory_[kIndexFrame]); out: the optimised poses. The views are host memory. Errors leave the map and
 * the previous resident frame untouched. Page-locked arrays in the plain layout are used in place, without a staging copy: raw_xyz as
 * rows of three doubles + timestamps as doubles (or override_timestamps) with order == NULL on the way in, all_world as rows of three
 * doubles on the way out; every other layout is staged through the handle's own pinned buffers. Same results. */
ctgn_status ctgn_frame_register(ctgn_handle h, ctgn_view raw_xyz, ctgn_view timestamps, size_t n, const uint32_t *order,
                                const ctgn_frame_options *fopts, double pose_io[14], const double t_begin_end[2],
                                const ctgn_options *opts, const ctgn_motion_prior *prior, const ctgn_robust_options *robust,
                                const ctgn_robust_prior *robust_prior, ctgn_frame_outputs *out, ctgn_summary *summary);
/* The same stages one by one, for a caller that keeps Odometry::DoRegister's own control flow around them — integration/odometry_gpu_arm.h
 * puts them behind the reference's InitializeFrame, TryRegister, undistortion loops and UpdateMap:
 *   ctgn_frame_begin         InitializeFrame (odometry.cpp:333-382): the scan is staged in processing order (`order` as above) and uploaded,
 *                            sub_sample_frame runs on it, and — fopts->sample_voxel_size being the keypoint voxel the first TryRegister will
 *                            ask for — the keypoint sampler behind it. out (may be NULL): sampled_indices, sampled_world_* = the sampled frame
 *                            under pose_initial (odometry.cpp:371-375), num_sampled, num_keypoints.
 *   ctgn_frame_try_register  TryRegister (odometry.cpp:525-601) on the resident sampled frame, any number of times (the robust retry loop,
 *                            :794-845): keypoints = grid_sampling at fopts->sample_voxel_size (sampled again only if it differs from the last
 *                            run's), the first max_num_keypoints of them; their world points from pose_io (the caller's current estimate, not
 *                            ctgn_frame_begin's); then ctgn_solve / ctgn_solve_robust. out: keypoint_indices, keypoint_world_*, num_keypoints.
 *   ctgn_frame_undistort     odometry.cpp:461-486 with the poses the host settled on (`pose` need not be what the registration returned):
 *                            out->all_world_* = every scan point, out->sampled_world_* = the sampled frame; the latter stays on the device as
 *                            the batch ctgn_frame_update_map inserts.
 * fopts->frame_voxel_size / override_timestamp(s) are read by ctgn_frame_begin only. Same error conventions as ctgn_frame_register. */
/* Optional first half of ctgn_frame_begin for a caller whose `order` takes time to make (std::shuffle of 132 k indices: 0.4-1 ms of one
 * host core): stages and uploads the scan in scan order and returns; the caller computes the order on ANOTHER thread meanwhile and hands it
 * to ctgn_frame_begin with raw_xyz.base == NULL and the same n ("the staged scan": the processing order is applied on the device either
 * way). fopts: override_timestamp(s) are read here; pose_initial / t_begin_end as for ctgn_frame_begin (which may pass others). */
ctgn_status ctgn_frame_stage(ctgn_handle h, ctgn_view raw_xyz, ctgn_view timestamps, size_t n, const ctgn_frame_options *fopts,
                             const double pose_initial[14], const double t_begin_end[2]);
ctgn_status ctgn_frame_begin(ctgn_handle h, ctgn_view raw_xyz, ctgn_view timestamps, size_t n, const uint32_t *order,
                             const ctgn_frame_options *fopts, const double pose_initial[14], const double t_begin_end[2],
                             ctgn_frame_outputs *out);
ctgn_status ctgn_frame_try_register(ctgn_handle h, const ctgn_frame_options *fopts, double pose_io[14], const double t_begin_end[2],
                                    const ctgn_options *opts, const ctgn_motion_prior *prior, const ctgn_robust_options *robust,
                                    const ctgn_robust_prior *robust_prior, ctgn_frame_outputs *out, ctgn_summary *summary);
ctgn_status ctgn_frame_undistort(ctgn_handle h, const double pose[14], const double t_begin_end[2], ctgn_frame_outputs *out);
/* Page-locked host memory for the caller's scan and output arrays: ctgn_frame_register / ctgn_frame / ctgn_frame_begin read x y z rows of
 * doubles + timestamps living in such memory in place, and ctgn_frame_register / ctgn_frame_undistort write `all_world` rows of three doubles
 * there in place (no staging copy either way). Plain hipHostMalloc / hipHostFree behind a C signature, for hosts that do
 * not link the HIP runtime themselves. */
ctgn_status ctgn_host_alloc(ctgn_handle h, size_t bytes, void **out);
ctgn_status ctgn_host_free(ctgn_handle h, void *p);
/* UpdateMap for the resident frame (odometry.cpp:936-952): RemoveElementsFarFromLocation(location, max_distance) on every level,
 * then — if add_points — insert the sampled frame's undistorted points. Needs the device-resident map (ctgn_map_set_update_mode 1).
 * inserted (host, num_sampled bytes, may be NULL): 1 where the point entered some level. */
ctgn_status ctgn_frame_update_map(ctgn_handle h, const double location[3], double max_distance, int32_t add_points,
                                  uint8_t *inserted);
/* ctgn_frame_register, then ctgn_frame_update_map(end translation, max_distance, success). */
ctgn_status ctgn_frame(ctgn_handle h, ctgn_view raw_xyz, ctgn_view timestamps, size_t n, const uint32_t *order,
                       const ctgn_frame_options *fopts, double pose_io[14], const double t_begin_end[2], const ctgn_options *opts,
                       const ctgn_motion_prior *prior, const ctgn_robust_options *robust, const ctgn_robust_prior *robust_prior,
                       double max_distance, ctgn_frame_outputs *out, ctgn_summary *summary);

/* -------------------------------------------------------------------------------------------------
 * Introspection for tests / measurement.
 * ---------------------------------------------------------------------------------------------- */
/* Per-keypoint outputs of the last accumulate pass (host copies; any pointer may be NULL):
 *   n_neighbors[i]; normal[3i..] (oriented, unit); a2d[i]; farthest[3i..] (the reference's
 *   `closest_point` = points[0]); used[i] (passed both gates). Requires ctgn_set_debug(h, 1). */
ctgn_status ctgn_set_debug(ctgn_handle h, int32_t enable);
ctgn_status ctgn_get_debug(ctgn_handle h, int32_t *n_neighbors, double *normal, double *a2d, double *farthest,
                           uint8_t *used, size_t n);
/* Host copy of the packed system of the last accumulate/solve. */
ctgn_status ctgn_get_system(ctgn_handle h, double out[CTGN_SYSTEM_DOUBLES]);
/* Bracket every accumulate launch with HIP events on the handle's stream (off by default). */
ctgn_status ctgn_set_profiling(ctgn_handle h, int32_t enable);
/* Average HIP-event time (ms) of the accumulate kernel over the launches that did work since the last reset. */
ctgn_status ctgn_kernel_timing(ctgn_handle h, double *avg_accumulate_ms, int32_t *launches, int32_t reset);
/* Keypoint ordering of the GN kernels: -1 = automatic (default), 0 = never, 1 = always. When ordered, the kernels work through
 * the upload in home-voxel order (positions sorted once per upload on the device, the kernels iterate on a position-ordered
 * working copy): same per-keypoint results, the packed sums then run in position order (a different, still fixed, rounding).
 * Automatic = when the searched map level exceeds the caches, or when the caller's iteration budget covers the ~80 us the sort
 * costs at 132 k keypoints (num_iters_icp >= 13 there); never below 32 k keypoints. Takes effect at the
 * next ctgn_set_keypoints. Debug capture and the robust route keep caller-order records and read through the order instead. */
ctgn_status ctgn_set_ordering(ctgn_handle h, int32_t mode);
/* The persistent small-frame kernel: 1 = solves of at most 1 024 keypoints on the default row kernel run as ONE launch — state init,
 * every GN iteration with an in-kernel barrier on one XCD, final re-transform; 0 (default) = always the three
 * launches per iteration. Off by default: once the partial sums were laid out block-major (the persistent kernel's exchange is where
 * the L2-channel conflict of the old layout was found) the three-launch loop runs a small frame's iteration in the same 28-29 us.
 * Same results up to the (fixed) order of the block sums. */
ctgn_status ctgn_set_persistent(ctgn_handle h, int32_t mode);
/* Neighbour pools: -1 = automatic (default: frames of at least 8 192 keypoints), 0 = never, 1 = always. With
 * pools a bounded search of a solve also keeps up to eight spare candidates behind the k neighbours and the radius inside which that
 * pool is complete; the next iterations first check the pool — if its k-th nearest member still lies inside that radius, shrunk by the
 * distance the keypoint has moved, the k nearest pool members ARE the neighbours — and search only the keypoints this does not cover.
 * Identical neighbour sets in identical order (bit-identical poses): only the time changes. */
ctgn_status ctgn_set_pools(ctgn_handle h, int32_t mode);
#ifdef __cplusplus
}
#endif
#endif /* CTGN_H */
