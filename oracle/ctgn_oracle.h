/*
 * ctgn_oracle.h — CPU restatement (plain C, double precision, dependency-free) of the Gauss–Newton
 * CT-ICP registration path of jedeschaud/ct_icp.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may build, load or call it. Nothing under ct_icp_amd/ links or imports it.
 *
 * PARITY PINNED (since round 2) against the reference's own sources: `make -C oracle _ref` compiles
 * src/ct_icp/{ct_icp,map,motion_model,cost_function,neighborhood_strategy}.cpp and the SlamCore sources they need,
 * where they lie under /root/reference, against header stand-ins for the absent libraries (oracle/shims/: mini-Eigen,
 * glog, tsl::robin_map, mini-Ceres) into oracle/_ref/libctgn_ref.so; tests/test_oracle_vs_ref.py holds every function
 * of this file to it (neighbour lists and ties bit-identical, poses to ~1e-16, both routes). The reference's own
 * tests hold no golden vectors for this path (TEST(CT_ICP, GN) is an empty body, test/unit/ct_icp/test_ct_icp.cxx:10-12).
 * Earlier anchors kept: (i) the property tests the reference does have (test_map.cxx:25-36,
 * test_neighborhood.cxx:40-53, test_cost_functions.cxx:70-105, test_types.cxx:20-31), re-expressed in
 * tests/test_oracle_*.py, (ii) an independent NumPy/SciPy re-derivation (oracle/numpy_check.py) and (iii) recovery
 * of a known ground-truth pose on noise-free synthetic planes. What the pin does not cover: the stand-ins themselves
 * (Eigen's JacobiSVD / LDLT and Ceres' trust-region loop are restated there too) — DESIGN.md section 12.
 *
 * Third-party arithmetic restated from published algorithms: Eigen 3 (unpinned `master` in the
 * reference's superbuild, superbuild/CMakeLists.txt:20-24) — Quaternion::slerp, Quaternion*Vector3,
 * toRotationMatrix, Quaternion(Matrix3), normalize, JacobiSVD on a symmetric 3x3 (== symmetric
 * eigen-decomposition), LDLT; libstdc++ std::priority_queue (push_heap / pop_heap).
 */
#ifndef CTGN_ORACLE_H
#define CTGN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_RESOLUTIONS 8
#define ORC_MAX_NEIGHBORS 64

typedef struct orc_map orc_map;

typedef struct {
    double resolution;
    double min_distance_between_points;
    int max_num_points;
} orc_resolution;

typedef struct {
    int num_iters_icp;
    int min_number_neighbors;
    int max_number_neighbors;
    int debug_print;
    double max_dist_to_plane_ct_icp;
    double threshold_orientation_norm;
} orc_options;

typedef struct {
    double beta_location_consistency;
    double beta_constant_velocity;
    double previous_begin_tr[3];
    double previous_end_tr[3];
} orc_motion_prior;

typedef struct {
    int success;
    int num_residuals_used;
    int num_iters;
    double last_step_norm;
    char error_log[256];
    /* phase split of the CPU baseline, seconds (mirrors the stopwatches at ct_icp.cpp:734-740) */
    double t_neighbors, t_normals, t_jacobian, t_solve, t_update;
} orc_summary;

/* ---- map (reference include/ct_icp/map.h) ---- */
orc_map *orc_map_create(const orc_resolution *res, int num_resolutions, double default_radius);
void orc_map_destroy(orc_map *m);
void orc_map_clear(orc_map *m);
/* InsertPointInVoxelMap for every resolution (map.h:196-206,261-293). inserted[i]=1 if kept anywhere. */
void orc_map_insert(orc_map *m, const double *xyz, size_t n, uint8_t *inserted);
/* RemoveElementsFarFromLocation (map.h:305-322). */
void orc_map_remove_far(orc_map *m, const double location[3], double distance);
uint64_t orc_map_num_points(const orc_map *m);               /* all resolutions */
uint64_t orc_map_num_voxels(const orc_map *m, int res_index);
uint64_t orc_map_export(const orc_map *m, int res_index, double *out_xyz, uint64_t capacity_points);
/* SearchParamsFromRadiusSearch (map.h:416-432). */
void orc_map_search_params(const orc_map *m, double radius, int *map_id, double *voxel_resolution,
                           int *voxel_neighborhood);
/* RadiusSearchInPlace (map.h:449-514), sensor_location == nullptr. Returns the neighbour count;
 * out_xyz is farthest-first. heap_mode 0 = libstdc++ priority_queue order (faithful), 1 = total order
 * (distance^2, visit index) — the GPU's documented tie rule. radius<=0 -> default_radius. */
int orc_map_radius_search(const orc_map *m, const double query[3], double radius, int max_num_neighbors,
                          int heap_mode, double *out_xyz);
/* Counting pass: voxels probed / hit and points scanned for one query. */
void orc_map_count(const orc_map *m, const double query[3], uint64_t *probed, uint64_t *hit, uint64_t *points);

/* ---- geometry helpers (Eigen semantics) ---- */
int orc_voxel_coord(double p, double voxel_size);                               /* types.cxx:15-17 */
double orc_alpha_timestamp(double t, double t_begin, double t_end);             /* types.h:192-219 */
void orc_quat_normalize(double q[4]);
void orc_quat_rotate(const double q[4], const double v[3], double out[3]);      /* Eigen q * v     */
void orc_quat_slerp(const double a[4], const double b[4], double t, double out[4]);
void orc_quat_to_matrix(const double q[4], double R[9]);                        /* row-major       */
void orc_matrix_to_quat(const double R[9], double q[4]);
/* TPose::InterpolatePose(...) * raw  (types.h:453-470 -> :360-366 -> :353-357). */
void orc_transform_point(const double pose[14], const double t_begin_end[2], double t, const double raw[3],
                         double out[3]);
void orc_transform_points(const double pose[14], const double t_begin_end[2], const double *t, const double *raw, size_t n,
                          double *out, int num_threads);                       /* odometry.cpp:461-486 */
/* TNeighborhood::ComputeNeighborhood(A2D|NORMAL) (neighborhood.h:225-257,285-316). Returns 0 if <5 pts. */
int orc_neighborhood(const double *pts_xyz, int n, double normal[3], double *a2d);
/* symmetric 3x3 eigen-decomposition, eigenvalues descending, V columns = eigenvectors (row-major 3x3) */
void orc_jacobi_svd3(const double C[9], double sv[3], double V[9]);
void orc_sym_eigen3(const double C[9], double evals[3], double V[9]);
/* pivoted LDL^T solve of a 12x12 symmetric system (Eigen A.ldlt().solve(b), ct_icp.cpp:914) */
void orc_ldlt_solve12(const double A[144], const double b[12], double x[12]);

/* ---- the hot path ---- */
/* One accumulation pass (ct_icp.cpp:746-857): A (12x12 row-major, NOT normalised), b, n_used.
 * Optional per-keypoint outputs (may be NULL): n_neighbors, normal(3), a2d, farthest(3), used.
 * num_threads<=1: serial, exactly the reference's order. >1: OpenMP over keypoints with per-thread A,b
 * summed in thread order (the CPU-N baseline of BASELINE.md). */
void orc_gn_accumulate(const orc_map *m, const double *raw_xyz, const double *world_xyz, const double *t,
                       size_t n, const double pose[14], const double t_begin_end[2], const orc_options *opts,
                       int heap_mode, int num_threads, double A[144], double b[12], int *n_used,
                       int32_t *n_neighbors, double *normal, double *a2d, double *farthest, uint8_t *used);
/* Normalise + motion prior + solve + pose update (ct_icp.cpp:877-962). Returns ||x||_2. */
double orc_gn_solve_update(double A[144], double b[12], int n_used, const orc_motion_prior *prior,
                           double pose[14], double x_out[12]);
/* DoRegisterGaussNewton (ct_icp.cpp:709-996). world_xyz is updated in place. Returns 0, or -5 when a
 * timestamp lies outside [t_begin, t_end] (the reference CHECK-aborts, types.h:456). */
int orc_register_gn(const orc_map *m, const double *raw_xyz, double *world_xyz, const double *t, size_t n,
                    double pose[14], const double t_begin_end[2], const orc_options *opts,
                    const orc_motion_prior *prior, int heap_mode, int num_threads, orc_summary *summary);

/* sub_sample_frame / grid_sampling (ct_icp.cpp:65-101): keeps the first point per voxel of the RAW
 * coordinates with `short` voxel indices; emits in first-insertion order (the reference's robin_map
 * iteration order is unspecified). Returns the number of kept indices. */
size_t orc_grid_sampling(const double *raw_xyz, size_t n, double voxel_size, uint32_t *out_indices);

/* AdaptiveSamplePointsInGrid (include/ct_icp/algorithm/sampling.h:55-110): range-banded grid sampling, first
 * num_points_per_voxel indices per voxel, at most max_num_points + 1 indices (sic, :96-106); order band, then voxel (z, y, x),
 * then index (the reference's std::unordered_map order is unspecified). (size_t) -1 on an invalid band list. */
size_t orc_adaptive_sampling(const double *raw_xyz, size_t n, int num_points_per_voxel, int max_num_points, int num_bands,
                             const double *distance, const double *voxel_size, uint32_t *out_indices);

/* ---- robust-loss (CERES-profile) route: DoRegisterCeres, ct_icp.cpp:457-707 (ctgn_oracle_robust.c) ---- */
enum { ORC_LOSS_STANDARD = 0, ORC_LOSS_CAUCHY = 1, ORC_LOSS_HUBER = 2, ORC_LOSS_TOLERANT = 3, ORC_LOSS_TRUNCATED = 4 };

typedef struct {                           /* CTICPOptions fields read by DoRegisterCeres (ct_icp.h:58-132) */
    int num_iters_icp, min_number_neighbors, max_number_neighbors, debug_print;
    int max_num_residuals, loss_function, ls_max_num_iters, num_closest_neighbors;
    double weight_alpha, weight_neighborhood, power_planarity, max_dist_to_plane_ct_icp;
    double ls_sigma, ls_tolerant_min_threshold;
    double threshold_orientation_norm, threshold_translation_norm;
} orc_robust_options;

typedef struct {                           /* PreviousFrameMotionModel (motion_model.h:42-58, motion_model.cpp:12-61) */
    double beta_location_consistency, beta_constant_velocity, beta_small_velocity, beta_orientation_consistency;
    double previous_begin_tr[3], previous_end_tr[3], previous_end_quat[4];
} orc_robust_prior;

typedef struct {
    double initial_cost, final_cost, final_radius;
    int iterations, num_successful_steps, num_unsuccessful_steps, termination;
} orc_lm_report;

void orc_loss_evaluate(int kind, double sigma, double tolerant_min, double s, double rho[3]);
/* CTFunctor<FunctorPointToPlane> residual and its 12 tangent partials [begin_quat | end_quat | begin_t | end_t]
 * (EigenQuaternionParameterization tangent), by forward-mode automatic differentiation. jac may be NULL. */
void orc_ct_point_to_plane(const double pose[14], double alpha, const double raw[3], const double ref[3],
                           const double normal[3], double weight, double *residual, double *jac);
size_t orc_robust_build(const orc_map *m, const double *raw_xyz, const double *world_xyz, const double *t, size_t n,
                        const double t_begin_end[2], const orc_robust_options *opts, int heap_mode, double *raw_out,
                        double *ref_out, double *normal_out, double *weight_out, double *alpha_out,
                        int32_t *keypoint_out);
double orc_robust_evaluate_fixed(const double *raw, const double *ref, const double *normal, const double *weight,
                                 const double *alpha, size_t n, const orc_robust_options *opts,
                                 const orc_robust_prior *prior, const double pose[14], double *H, double *g);
int orc_robust_solve_fixed(const double *raw, const double *ref, const double *normal, const double *weight,
                           const double *alpha, size_t n, const orc_robust_options *opts, const orc_robust_prior *prior,
                           double pose[14], int max_num_iterations, orc_lm_report *rep);
int orc_register_robust(const orc_map *m, const double *raw_xyz, double *world_xyz, const double *t, size_t n,
                        double pose[14], const double t_begin_end[2], const orc_robust_options *opts,
                        const orc_robust_prior *prior, int heap_mode, orc_summary *summary);

#ifdef __cplusplus
}
#endif
#endif
