// ct_icp_gpu.hpp — C++ host adapter over the C ABI of libctgn.so (include/ctgn.h).
//
// It mirrors, name for name, the part of the reference's C++ API that the GN path touches, so that reference-style
// caller code (Odometry::TryRegister, src/ct_icp/odometry.cpp:573-579) reads the same against this backend:
//
//   ct_icp::CT_ICP_Registration::Register(map, keypoints, frame, motion_model, strategy) -> ICPSummary
//                                                        (reference include/ct_icp/ct_icp.h:180-190)
//   ct_icp::ISlamMap subset: InsertPointCloud, RemoveElementsFarFromLocation, ClearMap, NumPoints, MapAsPointCloud,
//                            ComputeNeighborhoods / RadiusSearch      (reference include/ct_icp/map.h:14-83)
//   slam::WPoint3D (64-byte record), slam::Pose, ct_icp::TrajectoryFrame, CTICPOptions, ICPSummary,
//   PreviousFrameMotionModel (the fields GN reads).
//
// The header is Eigen-free (Eigen is not installed in this image): vectors are double[3] / double[4] with Eigen's
// memory layout (quaternion coeffs x,y,z,w), so inside the reference tree `Eigen::Map` views over the same bytes are
// free. INTEGRATION.md shows the ~40-line glue that makes this a `ct_icp::ISlamMap` + the `case GN:` hook.
#pragma once

#include <cstdint>
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ctgn.h"

namespace ct_icp_gpu {

// slam::WPoint3D (include/SlamCore/types.h:35-60): raw xyz @0, timestamp @24, world xyz @32, index_frame @56.
struct WPoint3D {
    double raw_point[3];
    double timestamp;
    double world_point[3];
    uint32_t index_frame = (uint32_t) -1;
    uint32_t _pad = 0;
};
static_assert(sizeof(WPoint3D) == 64, "WPoint3D must keep the reference's 64-byte layout");

// slam::TSE3<double> (include/SlamCore/types.h:104-138): Eigen::Quaterniond (coeffs x, y, z, w; 16-byte aligned) + Eigen::Vector3d.
struct alignas(16) SE3 {
    double quat[4] = {0, 0, 0, 1};
    double tr[3] = {0, 0, 0};
};
// slam::TPose<double> (types.h:161-167), the reference's member ORDER: pose, ref_timestamp, dest_timestamp, ref_frame_id, dest_frame_id
// (frame_id_t = unsigned int, types.h:19). Same size and offsets as the reference's struct as compiled (tests/test_oracle_vs_ref.py).
struct Pose {
    SE3 pose;
    double ref_timestamp = 0.0;
    double dest_timestamp = -1.0;
    unsigned int ref_frame_id = 0;
    unsigned int dest_frame_id = (unsigned int) -1;
};
static_assert(sizeof(SE3) == 64 && offsetof(SE3, tr) == 32, "slam::SE3 layout");
static_assert(sizeof(Pose) == 96 && offsetof(Pose, ref_timestamp) == 64 && offsetof(Pose, dest_timestamp) == 72 &&
              offsetof(Pose, ref_frame_id) == 80 && offsetof(Pose, dest_frame_id) == 84, "slam::Pose layout");

// ct_icp::TrajectoryFrame (include/ct_icp/types.h:31-61)
struct TrajectoryFrame {
    Pose begin_pose, end_pose;
    const double *BeginTr() const { return begin_pose.pose.tr; }
    const double *EndTr() const { return end_pose.pose.tr; }
    const double *BeginQuat() const { return begin_pose.pose.quat; }
    const double *EndQuat() const { return end_pose.pose.quat; }
};

enum CT_ICP_SOLVER { GN, CERES, ROBUST };                    // include/ct_icp/ct_icp.h:35-39

enum LEAST_SQUARES { STANDARD, CAUCHY, HUBER, TOLERANT, TRUNCATED };   // include/ct_icp/ct_icp.h:41-47

// The fields of ct_icp::CTICPOptions DoRegisterGaussNewton / DoRegisterCeres read (include/ct_icp/ct_icp.h:56-153),
// same defaults.
struct CTICPOptions {
    int num_iters_icp = 5;
    CT_ICP_SOLVER solver = CERES;
    int max_num_residuals = -1;
    double weight_alpha = 0.9;
    double weight_neighborhood = 0.1;
    double power_planarity = 2.0;
    int max_number_neighbors = 20;
    int min_number_neighbors = 20;
    int num_closest_neighbors = 1;
    double threshold_orientation_norm = 0.0001;
    double threshold_translation_norm = 0.001;
    LEAST_SQUARES loss_function = CAUCHY;
    int ls_max_num_iters = 1;
    double ls_sigma = 0.1;
    double ls_tolerant_min_threshold = 0.05;
    double max_dist_to_plane_ct_icp = 0.3;
    bool debug_print = true;
};

// ct_icp::ICPSummary (include/ct_icp/ct_icp.h:155-169)
struct ICPSummary {
    bool success = false;
    int num_residuals_used = 0;
    int num_iters = 0;
    std::string error_log;
    double duration_total = 0., duration_init = 0., avg_duration_iter = 0., avg_duration_neighborhood = 0.,
           avg_duration_solve = 0.;
};

// The part of ct_icp::PreviousFrameMotionModel the solvers read (GN: src/ct_icp/ct_icp.cpp:888-908; CERES:
// src/ct_icp/motion_model.cpp:12-61).
struct PreviousFrameMotionModel {
    struct Options {
        double beta_location_consistency = 0.001;
        double beta_constant_velocity = 0.001;
        double beta_small_velocity = 0.0;
        double beta_orientation_consistency = 0.0;
    } options;
    TrajectoryFrame previous_frame;
    const TrajectoryFrame &PreviousFrame() const { return previous_frame; }
    void UpdateState(const TrajectoryFrame &optimized, int /*frame_index*/) { previous_frame = optimized; }
};

struct Neighborhood {            // slam::Neighborhood::points, farthest first (include/ct_icp/map.h:508-513)
    std::vector<double> points;  // 3 * n
    size_t size() const { return points.size() / 3; }
};

// "GPU_VOXEL_HASHMAP": the ISlamMap implementation that keeps the voxel map resident on the GPU.
class GpuVoxelMap {
public:
    struct ResolutionParam {
        double resolution = 0.5, min_distance_between_points = 0.1;
        int max_num_points = 40;
    };
    struct Options {                                          // MultipleResolutionVoxelMap::Options (map.h:115-133)
        std::vector<ResolutionParam> resolutions = {{0.2, 0.03, 50}, {0.5, 0.1, 40}, {1.5, 0.15, 40}};
        double default_radius = 0.8;
        int device = 0;
        static std::string Type() { return "GPU_VOXEL_HASHMAP"; }
    };

    GpuVoxelMap() : GpuVoxelMap(Options()) {}
    explicit GpuVoxelMap(const Options &options) : options_(options) {
        ctgn_map_options mo;
        ctgn_map_options_default(&mo);
        mo.num_resolutions = (int32_t) options.resolutions.size();
        mo.device = options.device;
        mo.default_radius = options.default_radius;
        for (size_t i = 0; i < options.resolutions.size() && i < CTGN_MAX_RESOLUTIONS; ++i)
            mo.resolutions[i] = ctgn_resolution_param{options.resolutions[i].resolution,
                                                      options.resolutions[i].min_distance_between_points,
                                                      options.resolutions[i].max_num_points, 0};
        ctgn_status st = ctgn_create(&mo, &h_);
        if (st != CTGN_OK) throw std::runtime_error(std::string("ctgn_create: ") + ctgn_status_string(st));
    }
    ~GpuVoxelMap() { ctgn_destroy(h_); }
    GpuVoxelMap(const GpuVoxelMap &) = delete;
    GpuVoxelMap &operator=(const GpuVoxelMap &) = delete;

    // InsertPointCloud(pointcloud, out_selected_points) on the world points of a frame (map.h:296-300 -> :153-254)
    void InsertPointCloud(const std::vector<WPoint3D> &frame, std::vector<size_t> &out_selected_points) {
        std::vector<uint8_t> kept(frame.size());
        check(ctgn_map_insert(h_, frame.empty() ? nullptr : frame[0].world_point, sizeof(WPoint3D), CTGN_F64, frame.size(),
                              kept.data()));
        out_selected_points.clear();
        for (size_t i = 0; i < kept.size(); ++i)
            if (kept[i]) out_selected_points.push_back(i);
    }
    void InsertPoints(const double *xyz, size_t n, size_t stride_bytes = 24) {
        check(ctgn_map_insert(h_, xyz, stride_bytes, CTGN_F64, n, nullptr));
    }
    void RemoveElementsFarFromLocation(const double location[3], double distance) {
        check(ctgn_map_remove_far(h_, location, distance));
    }
    void ClearMap() { check(ctgn_map_clear(h_)); }
    size_t NumPoints() const {
        uint64_t n = 0;
        ctgn_map_num_points(h_, &n);
        return (size_t) n;
    }
    std::vector<double> MapAsPointCloud(int resolution_index = 0) const {
        uint64_t n = 0;
        ctgn_map_export(h_, resolution_index, nullptr, 0, &n);
        std::vector<double> out(3 * n);
        ctgn_map_export(h_, resolution_index, out.data(), n, &n);
        return out;
    }
    // ComputeNeighborhoods(queries, max_num_neighbors) (map.h:532-541), batched on the GPU
    std::vector<Neighborhood> ComputeNeighborhoods(const std::vector<double> &queries_xyz, int max_num_neighbors,
                                                   double radius = -1.0) {
        const size_t n = queries_xyz.size() / 3;
        std::vector<double> out(n * (size_t) max_num_neighbors * 3);
        std::vector<int32_t> cnt(n);
        check(ctgn_map_radius_search(h_, queries_xyz.data(), n, radius, max_num_neighbors, out.data(), cnt.data()));
        std::vector<Neighborhood> res(n);
        for (size_t i = 0; i < n; ++i)
            res[i].points.assign(out.begin() + i * max_num_neighbors * 3, out.begin() + (i * max_num_neighbors + cnt[i]) * 3);
        return res;
    }
    Neighborhood RadiusSearch(const double query[3], double radius, int max_num_neighbors) {
        return ComputeNeighborhoods(std::vector<double>(query, query + 3), max_num_neighbors, radius)[0];
    }
    // the in-place spellings of ISlamMap (include/ct_icp/map.h:47-82)
    void RadiusSearchInPlace(const double query[3], Neighborhood &neighborhood, double radius, int max_num_neighbors) {
        neighborhood = RadiusSearch(query, radius, max_num_neighbors);
    }
    void ComputeNeighborhoodInPlace(const double query[3], int max_num_neighbors, Neighborhood &neighborhood) {
        neighborhood = RadiusSearch(query, -1.0 /* default_radius, map.h:527-530 */, max_num_neighbors);
    }
    // InsertPointCloud(pointcloud, frame_poses, out_indices) for a frame that only carries RAW points (map.h:153-184): the world
    // points are derived from the begin / end pose first (continuous-time interpolation by timestamp), on the GPU, then inserted.
    void InsertPointCloud(std::vector<WPoint3D> &frame, const Pose &begin_pose, const Pose &end_pose, std::vector<size_t> &out_selected_points) {
        if (!frame.empty()) {
            double pose[14];
            std::memcpy(pose, begin_pose.pose.quat, 32); std::memcpy(pose + 4, begin_pose.pose.tr, 24);
            std::memcpy(pose + 7, end_pose.pose.quat, 32); std::memcpy(pose + 11, end_pose.pose.tr, 24);
            const double tbe[2] = {begin_pose.dest_timestamp, end_pose.dest_timestamp};
            ctgn_view raw{frame[0].raw_point, sizeof(WPoint3D), CTGN_F64, 0};
            ctgn_view ts{&frame[0].timestamp, sizeof(WPoint3D), CTGN_F64, 0};
            check(ctgn_transform_points(h_, raw, ts, frame.size(), pose, tbe, frame[0].world_point, sizeof(WPoint3D), CTGN_F64));
        }
        InsertPointCloud(frame, out_selected_points);
    }

    ctgn_handle handle() const { return h_; }
    const Options &GetOptions() const { return options_; }

private:
    void check(ctgn_status st) const {
        if (st != CTGN_OK) throw std::runtime_error(std::string("libctgn: ") + ctgn_last_error(h_));
    }
    Options options_;
    ctgn_handle h_ = nullptr;
};

// ct_icp::CT_ICP_Registration for `solver: GN` and `solver: CERES` (CONTINUOUS_TIME, POINT_TO_PLANE).
class CT_ICP_Registration {
public:
    CTICPOptions &Options() { return options_; }
    const CTICPOptions &Options() const { return options_; }

    // Register(voxel_map, keypoints, trajectory_frame, motion_model, strategy) — the vector<slam::WPoint3D> overload
    // (src/ct_icp/ct_icp.cpp:1026-1037). Updates the frame's poses and the keypoints' world points in place.
    ICPSummary Register(GpuVoxelMap &voxel_map, std::vector<WPoint3D> &keypoints, TrajectoryFrame &trajectory_frame,
                        const PreviousFrameMotionModel *motion_model = nullptr, void * /*strategy: unused by GN*/ = nullptr) {
        if (options_.solver == ROBUST) throw std::runtime_error("Unsupported Solver Type");  // ct_icp.cpp:1022
        if (options_.solver == CERES) return RegisterCeres(voxel_map, keypoints, trajectory_frame, motion_model);
        ctgn_options o;
        ctgn_options_default(&o);
        o.num_iters_icp = options_.num_iters_icp;
        o.min_number_neighbors = options_.min_number_neighbors;
        o.max_number_neighbors = options_.max_number_neighbors;
        o.debug_print = options_.debug_print ? 1 : 0;
        o.max_dist_to_plane_ct_icp = options_.max_dist_to_plane_ct_icp;
        o.threshold_orientation_norm = options_.threshold_orientation_norm;
        ctgn_motion_prior prior, *pp = nullptr;
        if (motion_model) {                                                                  // ct_icp.cpp:885-889
            prior.beta_location_consistency = motion_model->options.beta_location_consistency;
            prior.beta_constant_velocity = motion_model->options.beta_constant_velocity;
            std::memcpy(prior.previous_begin_tr, motion_model->PreviousFrame().BeginTr(), 24);
            std::memcpy(prior.previous_end_tr, motion_model->PreviousFrame().EndTr(), 24);
            pp = &prior;
        }
        double pose[14];
        std::memcpy(pose, trajectory_frame.begin_pose.pose.quat, 32);
        std::memcpy(pose + 4, trajectory_frame.begin_pose.pose.tr, 24);
        std::memcpy(pose + 7, trajectory_frame.end_pose.pose.quat, 32);
        std::memcpy(pose + 11, trajectory_frame.end_pose.pose.tr, 24);
        const double tbe[2] = {trajectory_frame.begin_pose.dest_timestamp, trajectory_frame.end_pose.dest_timestamp};
        const size_t n = keypoints.size();
        WPoint3D dummy{};
        WPoint3D *base = n ? keypoints.data() : &dummy;
        ctgn_view raw{base->raw_point, sizeof(WPoint3D), CTGN_F64, 0};
        ctgn_view ts{&base->timestamp, sizeof(WPoint3D), CTGN_F64, 0};
        ctgn_summary s;
        ctgn_status st = ctgn_register(voxel_map.handle(), raw, base->world_point, sizeof(WPoint3D), CTGN_F64, ts, n, pose, tbe,
                                       &o, pp, &s);
        ICPSummary out;
        if (st != CTGN_OK) {                 // hard errors of the reference (CHECK aborts) and HIP errors: soft failure
            out.success = false;
            out.error_log = ctgn_last_error(voxel_map.handle());
            return out;
        }
        std::memcpy(trajectory_frame.begin_pose.pose.quat, pose, 32);
        std::memcpy(trajectory_frame.begin_pose.pose.tr, pose + 4, 24);
        std::memcpy(trajectory_frame.end_pose.pose.quat, pose + 7, 32);
        std::memcpy(trajectory_frame.end_pose.pose.tr, pose + 11, 24);
        out.success = s.success != 0;
        out.num_residuals_used = s.num_residuals_used;
        out.num_iters = s.num_iters;
        out.error_log = s.error_log;
        out.duration_total = s.duration_total_ms;                       // milliseconds, as the reference's duration_ms (ct_icp.cpp:24-27)
        out.duration_init = s.duration_init_ms;
        out.avg_duration_iter = s.avg_duration_iter_ms;
        out.avg_duration_neighborhood = s.avg_duration_neighborhood_ms;
        out.avg_duration_solve = s.avg_duration_solve_ms;
        return out;
    }

private:
    // `case CERES:` (src/ct_icp/ct_icp.cpp:1003-1007 -> DoRegisterCeres :457-707)
    ICPSummary RegisterCeres(GpuVoxelMap &voxel_map, std::vector<WPoint3D> &keypoints, TrajectoryFrame &trajectory_frame,
                             const PreviousFrameMotionModel *motion_model) {
        ctgn_robust_options o;
        ctgn_robust_options_default(&o);
        o.num_iters_icp = options_.num_iters_icp;
        o.min_number_neighbors = options_.min_number_neighbors;
        o.max_number_neighbors = options_.max_number_neighbors;
        o.debug_print = options_.debug_print ? 1 : 0;
        o.max_num_residuals = options_.max_num_residuals;
        o.loss_function = (int32_t) options_.loss_function;
        o.ls_max_num_iters = options_.ls_max_num_iters;
        o.num_closest_neighbors = options_.num_closest_neighbors;
        o.weight_alpha = options_.weight_alpha;
        o.weight_neighborhood = options_.weight_neighborhood;
        o.power_planarity = options_.power_planarity;
        o.max_dist_to_plane_ct_icp = options_.max_dist_to_plane_ct_icp;
        o.ls_sigma = options_.ls_sigma;
        o.ls_tolerant_min_threshold = options_.ls_tolerant_min_threshold;
        o.threshold_orientation_norm = options_.threshold_orientation_norm;
        o.threshold_translation_norm = options_.threshold_translation_norm;
        ctgn_robust_prior prior, *pp = nullptr;
        if (motion_model) {                                                                  // ct_icp.cpp:608-610
            prior.beta_location_consistency = motion_model->options.beta_location_consistency;
            prior.beta_constant_velocity = motion_model->options.beta_constant_velocity;
            prior.beta_small_velocity = motion_model->options.beta_small_velocity;
            prior.beta_orientation_consistency = motion_model->options.beta_orientation_consistency;
            std::memcpy(prior.previous_begin_tr, motion_model->PreviousFrame().BeginTr(), 24);
            std::memcpy(prior.previous_end_tr, motion_model->PreviousFrame().EndTr(), 24);
            std::memcpy(prior.previous_end_quat, motion_model->PreviousFrame().EndQuat(), 32);
            pp = &prior;
        }
        double pose[14];
        std::memcpy(pose, trajectory_frame.begin_pose.pose.quat, 32);
        std::memcpy(pose + 4, trajectory_frame.begin_pose.pose.tr, 24);
        std::memcpy(pose + 7, trajectory_frame.end_pose.pose.quat, 32);
        std::memcpy(pose + 11, trajectory_frame.end_pose.pose.tr, 24);
        const double tbe[2] = {trajectory_frame.begin_pose.dest_timestamp, trajectory_frame.end_pose.dest_timestamp};
        const size_t n = keypoints.size();
        WPoint3D dummy{};
        WPoint3D *base = n ? keypoints.data() : &dummy;
        ctgn_view raw{base->raw_point, sizeof(WPoint3D), CTGN_F64, 0};
        ctgn_view ts{&base->timestamp, sizeof(WPoint3D), CTGN_F64, 0};
        ctgn_summary s;
        ctgn_status st = ctgn_register_robust(voxel_map.handle(), raw, base->world_point, sizeof(WPoint3D), CTGN_F64, ts, n, pose,
                                              tbe, &o, pp, &s);
        if (st == CTGN_ERR_SOLVER) throw std::runtime_error("Error During Optimization");    // ct_icp.cpp:628-631
        ICPSummary out;
        if (st != CTGN_OK) {
            out.success = false;
            out.error_log = ctgn_last_error(voxel_map.handle());
            return out;
        }
        std::memcpy(trajectory_frame.begin_pose.pose.quat, pose, 32);
        std::memcpy(trajectory_frame.begin_pose.pose.tr, pose + 4, 24);
        std::memcpy(trajectory_frame.end_pose.pose.quat, pose + 7, 32);
        std::memcpy(trajectory_frame.end_pose.pose.tr, pose + 11, 24);
        out.success = s.success != 0;
        out.num_residuals_used = s.num_residuals_used;
        out.num_iters = s.num_iters;
        out.error_log = s.error_log;
        out.duration_total = s.duration_total_ms;                       // milliseconds, as the reference's duration_ms (ct_icp.cpp:24-27)
        out.duration_init = s.duration_init_ms;
        out.avg_duration_iter = s.avg_duration_iter_ms;
        out.avg_duration_neighborhood = s.avg_duration_neighborhood_ms;
        out.avg_duration_solve = s.avg_duration_solve_ms;
        return out;
    }

    CTICPOptions options_;
};

// ---- the steps either side of the path, same names as the reference's free functions (the map argument carries the device handle)

// ct_icp::grid_sampling (src/ct_icp/ct_icp.cpp:86-101): keypoints = the first point of every voxel of the RAW coordinates
inline void grid_sampling(GpuVoxelMap &voxel_map, const std::vector<WPoint3D> &frame, std::vector<WPoint3D> &keypoints,
                          double size_voxel_subsampling) {
    keypoints.clear();
    if (frame.empty()) return;
    std::vector<uint32_t> idx(frame.size());
    size_t count = 0;
    ctgn_view xyz{const_cast<double *>(frame[0].raw_point), sizeof(WPoint3D), CTGN_F64, 0};
    ctgn_status st = ctgn_grid_sampling(voxel_map.handle(), xyz, frame.size(), size_voxel_subsampling, idx.data(), &count);
    if (st != CTGN_OK) throw std::runtime_error(std::string("ctgn: ") + ctgn_last_error(voxel_map.handle()));
    keypoints.reserve(count);
    for (size_t k = 0; k < count; ++k) keypoints.push_back(frame[idx[k]]);
}

// ct_icp::sub_sample_frame (src/ct_icp/ct_icp.cpp:65-83): in-place variant
inline void sub_sample_frame(GpuVoxelMap &voxel_map, std::vector<WPoint3D> &frame, double size_voxel) {
    std::vector<WPoint3D> kept;
    grid_sampling(voxel_map, frame, kept, size_voxel);
    frame.swap(kept);
}

// ct_icp::AdaptiveGridSamplingOptions / AdaptiveSamplePointsInGrid (include/ct_icp/algorithm/sampling.h:13-26,55-110)
struct AdaptiveGridSamplingOptions {
    int num_points_per_voxel = 1;
    int max_num_points = -1;
    std::vector<std::pair<double, double>> distance_voxel_size = {{0.5, 0.1}, {2.0, 0.2}, {4., 0.4}, {8., 0.8}, {16., 1.6}, {200., -1.}};
};
inline std::vector<size_t> AdaptiveSamplePointsInGrid(GpuVoxelMap &voxel_map, const std::vector<WPoint3D> &frame,
                                                      const AdaptiveGridSamplingOptions &options) {
    std::vector<size_t> indices;
    if (frame.empty()) return indices;
    ctgn_adaptive_sampling_options o;
    ctgn_adaptive_sampling_options_default(&o);
    if (options.distance_voxel_size.size() > CTGN_ADAPTIVE_MAX_BANDS) throw std::runtime_error("ctgn: too many sampling bands");
    o.num_points_per_voxel = options.num_points_per_voxel;
    o.max_num_points = options.max_num_points;
    o.num_bands = (int32_t) options.distance_voxel_size.size();
    for (size_t j = 0; j < options.distance_voxel_size.size(); ++j) {
        o.distance[j] = options.distance_voxel_size[j].first;
        o.voxel_size[j] = options.distance_voxel_size[j].second;
    }
    std::vector<uint32_t> idx(frame.size());
    size_t count = 0;
    ctgn_view xyz{const_cast<double *>(frame[0].raw_point), sizeof(WPoint3D), CTGN_F64, 0};
    ctgn_status st = ctgn_adaptive_sampling(voxel_map.handle(), xyz, frame.size(), &o, idx.data(), &count);
    if (st != CTGN_OK) throw std::runtime_error(std::string("ctgn: ") + ctgn_last_error(voxel_map.handle()));
    indices.assign(idx.begin(), idx.begin() + count);
    return indices;
}

// the undistortion loop of Odometry::DoRegister (src/ct_icp/odometry.cpp:461-486): world_point = InterpolatePose(t) * raw_point
inline void TransformFrame(GpuVoxelMap &voxel_map, std::vector<WPoint3D> &frame, const TrajectoryFrame &trajectory_frame) {
    if (frame.empty()) return;
    double pose[14];
    std::memcpy(pose, trajectory_frame.begin_pose.pose.quat, 32);
    std::memcpy(pose + 4, trajectory_frame.begin_pose.pose.tr, 24);
    std::memcpy(pose + 7, trajectory_frame.end_pose.pose.quat, 32);
    std::memcpy(pose + 11, trajectory_frame.end_pose.pose.tr, 24);
    const double tbe[2] = {trajectory_frame.begin_pose.dest_timestamp, trajectory_frame.end_pose.dest_timestamp};
    ctgn_view raw{frame[0].raw_point, sizeof(WPoint3D), CTGN_F64, 0};
    ctgn_view ts{&frame[0].timestamp, sizeof(WPoint3D), CTGN_F64, 0};
    ctgn_status st = ctgn_transform_points(voxel_map.handle(), raw, ts, frame.size(), pose, tbe, frame[0].world_point, sizeof(WPoint3D),
                                           CTGN_F64);
    if (st != CTGN_OK) throw std::runtime_error(std::string("ctgn: ") + ctgn_last_error(voxel_map.handle()));
}

// One frame of Odometry::DoRegister with the scan resident on the device (ctgn_frame_register / ctgn_frame_update_map, SURVEY.md
// section 8f): InitializeFrame's sub_sample_frame + initial transform (src/ct_icp/odometry.cpp:333-382), TryRegister's grid_sampling +
// Register (:526-590) and both undistortion loops (:461-486) in one call — `all_corrected_points` receives every scan point with its
// world point, `corrected_points` the sampled frame, `keypoint_indices` the scan indices of the keypoints — then UpdateMap's evict +
// insert (:936-952) from the device-resident corrected points, after the host has decided (add_points). The map must maintain itself
// on the device: ctgn_map_set_update_mode(map.handle(), 1) right after construction. GN route (options.solver == GN).
struct FrameOptions {
    double voxel_size = 0.5;                 // OdometryOptions::voxel_size (sub_sample_frame)
    double sample_voxel_size = 1.5;          // OdometryOptions::sample_voxel_size (grid_sampling); <= 0: sampling NONE
    int max_num_keypoints = -1;
    const std::vector<uint32_t> *order = nullptr;    // the caller's shuffle of the scan (odometry.cpp:349), or scan order
    uint64_t shuffle_seed = 0;               // != 0 (and no `order`): the shuffle is made on the device (ctgn_frame_options::shuffle_seed)
};
inline ICPSummary RegisterFrame(GpuVoxelMap &voxel_map, const CTICPOptions &options, const FrameOptions &frame_options,
                                const std::vector<WPoint3D> &scan, TrajectoryFrame &trajectory_frame,
                                const PreviousFrameMotionModel *motion_model, std::vector<WPoint3D> *all_corrected_points,
                                std::vector<WPoint3D> *corrected_points, std::vector<uint32_t> *keypoint_indices) {
    if (options.solver != GN) throw std::runtime_error("RegisterFrame: GN route only (use ctgn_frame_register for the robust route)");
    ctgn_options o;
    ctgn_options_default(&o);
    o.num_iters_icp = options.num_iters_icp;
    o.min_number_neighbors = options.min_number_neighbors;
    o.max_number_neighbors = options.max_number_neighbors;
    o.debug_print = options.debug_print ? 1 : 0;
    o.max_dist_to_plane_ct_icp = options.max_dist_to_plane_ct_icp;
    o.threshold_orientation_norm = options.threshold_orientation_norm;
    ctgn_motion_prior prior, *pp = nullptr;
    if (motion_model) {
        prior.beta_location_consistency = motion_model->options.beta_location_consistency;
        prior.beta_constant_velocity = motion_model->options.beta_constant_velocity;
        std::memcpy(prior.previous_begin_tr, motion_model->PreviousFrame().BeginTr(), 24);
        std::memcpy(prior.previous_end_tr, motion_model->PreviousFrame().EndTr(), 24);
        pp = &prior;
    }
    ctgn_frame_options fo;
    ctgn_frame_options_default(&fo);
    fo.frame_voxel_size = frame_options.voxel_size;
    fo.sample_voxel_size = frame_options.sample_voxel_size;
    fo.max_num_keypoints = frame_options.max_num_keypoints;
    fo.shuffle_seed = frame_options.shuffle_seed;
    double pose[14];
    std::memcpy(pose, trajectory_frame.begin_pose.pose.quat, 32);
    std::memcpy(pose + 4, trajectory_frame.begin_pose.pose.tr, 24);
    std::memcpy(pose + 7, trajectory_frame.end_pose.pose.quat, 32);
    std::memcpy(pose + 11, trajectory_frame.end_pose.pose.tr, 24);
    const double tbe[2] = {trajectory_frame.begin_pose.dest_timestamp, trajectory_frame.end_pose.dest_timestamp};
    const size_t n = scan.size();
    WPoint3D dummy{};
    const WPoint3D *base = n ? scan.data() : &dummy;
    ctgn_view raw{base->raw_point, sizeof(WPoint3D), CTGN_F64, 0};
    ctgn_view ts{&base->timestamp, sizeof(WPoint3D), CTGN_F64, 0};
    ctgn_frame_outputs fout{};
    std::vector<uint32_t> sampled(n), kps(n);
    std::vector<double> sampled_world(3 * n);
    if (all_corrected_points) {
        *all_corrected_points = scan;                                        // raw point, timestamp, index_frame (odometry.cpp:472-476)
        fout.all_world_base = n ? (*all_corrected_points)[0].world_point : nullptr;
        fout.all_world_stride_bytes = sizeof(WPoint3D);
        fout.all_world_dtype = CTGN_F64;
    }
    fout.sampled_indices = sampled.data();
    fout.keypoint_indices = kps.data();
    fout.sampled_world_base = sampled_world.data();
    fout.sampled_world_stride_bytes = 24;
    fout.sampled_world_dtype = CTGN_F64;
    ctgn_summary s;
    const ctgn_status st = ctgn_frame_register(voxel_map.handle(), raw, ts, n, frame_options.order ? frame_options.order->data() : nullptr, &fo,
                                               pose, tbe, &o, pp, nullptr, nullptr, &fout, &s);
    ICPSummary out;
    if (st != CTGN_OK) {
        out.success = false;
        out.error_log = ctgn_last_error(voxel_map.handle());
        return out;
    }
    std::memcpy(trajectory_frame.begin_pose.pose.quat, pose, 32);
    std::memcpy(trajectory_frame.begin_pose.pose.tr, pose + 4, 24);
    std::memcpy(trajectory_frame.end_pose.pose.quat, pose + 7, 32);
    std::memcpy(trajectory_frame.end_pose.pose.tr, pose + 11, 24);
    if (corrected_points) {
        corrected_points->resize(fout.num_sampled);
        for (size_t k = 0; k < fout.num_sampled; ++k) {
            (*corrected_points)[k] = scan[sampled[k]];
            std::memcpy((*corrected_points)[k].world_point, &sampled_world[3 * k], 24);
        }
    }
    if (keypoint_indices) keypoint_indices->assign(kps.begin(), kps.begin() + fout.num_keypoints);
    out.success = s.success != 0;
    out.num_residuals_used = s.num_residuals_used;
    out.num_iters = s.num_iters;
    out.error_log = s.error_log;
    out.duration_total = s.duration_total_ms;
    out.duration_init = s.duration_init_ms;
    out.avg_duration_iter = s.avg_duration_iter_ms;
    out.avg_duration_neighborhood = s.avg_duration_neighborhood_ms;
    out.avg_duration_solve = s.avg_duration_solve_ms;
    return out;
}

// Odometry::UpdateMap's map part for the frame RegisterFrame left on the device (odometry.cpp:936-952)
inline void UpdateMapFromFrame(GpuVoxelMap &voxel_map, const double location[3], double max_distance, bool add_points) {
    const ctgn_status st = ctgn_frame_update_map(voxel_map.handle(), location, max_distance, add_points ? 1 : 0, nullptr);
    if (st != CTGN_OK) throw std::runtime_error(std::string("ctgn: ") + ctgn_last_error(voxel_map.handle()));
}

}  // namespace ct_icp_gpu
