// integration/gn_gpu_arm.h — the GPU arms of the two solver routes. A maintainer adds ONE statement at the top of each function in
// src/ct_icp/ct_icp.cpp (after #include "ct_icp/gn_gpu_arm.h"):
//
//   ICPSummary CT_ICP_Registration::DoRegisterGaussNewton(...) {                                   // :709
//       if (auto gpu = GpuGaussNewton(voxels_map, raw_kpts, world_kpts, timestamps, frame_to_optimize, motion_model, Options())) return *gpu;
//   ICPSummary CT_ICP_Registration::DoRegisterCeres(...) {                                         // :457
//       if (auto gpu = GpuCeres(voxels_map, raw_kpts, world_kpts, timestamps, frame_to_optimize, _previous_frame, Options())) return *gpu;
//
// Both return std::nullopt — and the reference's own CPU loop runs, untouched — when the map is not a GpuVoxelMap, when a view's
// source type is not FLOAT32 / FLOAT64, or (CERES) when the configuration is not CONTINUOUS_TIME + POINT_TO_PLANE.
// oracle/Makefile applies exactly these two insertions to the reference's ct_icp.cpp on its way into the compiler and links the result with
// libctgn.so (target _ref/glue_check); tests/test_integration_glue.py compiles this header verbatim.
#ifndef CT_ICP_GN_GPU_ARM_H
#define CT_ICP_GN_GPU_ARM_H

#include <optional>

#include <ct_icp/ct_icp.h>
#include <ct_icp/gpu_map.h>

namespace ct_icp {

    namespace ctgn_glue {
        // slam::SE3::Parameters() order (types.h:378-385): qx qy qz qw tx ty tz
        inline void pose_to_array(const TrajectoryFrame &frame, double pose[14], double tbe[2]) {
            for (int i = 0; i < 7; ++i) {
                pose[i] = frame.begin_pose.pose[i];
                pose[7 + i] = frame.end_pose.pose[i];
            }
            tbe[0] = frame.begin_pose.dest_timestamp;
            tbe[1] = frame.end_pose.dest_timestamp;
        }

        inline void array_to_pose(const double pose[14], TrajectoryFrame &frame) {           // in place, as ct_icp.cpp:950-962
            frame.begin_pose.QuatRef() = Eigen::Quaterniond(pose[3], pose[0], pose[1], pose[2]);
            frame.begin_pose.TrRef() = Eigen::Vector3d(pose[4], pose[5], pose[6]);
            frame.end_pose.QuatRef() = Eigen::Quaterniond(pose[10], pose[7], pose[8], pose[9]);
            frame.end_pose.TrRef() = Eigen::Vector3d(pose[11], pose[12], pose[13]);
        }

        inline ICPSummary to_summary(ctgn_status st, const ctgn_summary &s, ctgn_handle h) {
            ICPSummary summary;
            summary.success = (st == CTGN_OK) && s.success;                                  // ct_icp.cpp:869 / :992
            summary.num_residuals_used = s.num_residuals_used;                               // :993
            summary.num_iters = s.num_iters;
            summary.error_log = (st == CTGN_OK) ? std::string(s.error_log) : std::string(ctgn_last_error(h));
            summary.duration_total = s.duration_total_ms;                                    // ct_icp.h:164-168, milliseconds
            summary.duration_init = s.duration_init_ms;
            summary.avg_duration_iter = s.avg_duration_iter_ms;
            summary.avg_duration_neighborhood = s.avg_duration_neighborhood_ms;
            summary.avg_duration_solve = s.avg_duration_solve_ms;
            return summary;
        }

        // the CTICPOptions fields DoRegisterGaussNewton reads (ct_icp.cpp:727,743,762,803,866,978)
        inline void gn_options_of(const CTICPOptions &options, ctgn_options *co) {
            ctgn_options_default(co);
            co->num_iters_icp = options.num_iters_icp;
            co->min_number_neighbors = options.min_number_neighbors;
            co->max_number_neighbors = options.max_number_neighbors;
            co->debug_print = options.debug_print ? 1 : 0;
            co->max_dist_to_plane_ct_icp = options.max_dist_to_plane_ct_icp;
            co->threshold_orientation_norm = options.threshold_orientation_norm;
        }

        inline void gn_prior_of(const PreviousFrameMotionModel &model, ctgn_motion_prior *prior) {       // ct_icp.cpp:885-908
            prior->beta_location_consistency = model.GetOptionsConst().beta_location_consistency;
            prior->beta_constant_velocity = model.GetOptionsConst().beta_constant_velocity;
            for (int c = 0; c < 3; ++c) {
                prior->previous_begin_tr[c] = model.PreviousFrame().BeginTr()[c];
                prior->previous_end_tr[c] = model.PreviousFrame().EndTr()[c];
            }
        }

        // the CTICPOptions fields DoRegisterCeres reads (ct_icp.h:58-132)
        inline void robust_options_of(const CTICPOptions &options, ctgn_robust_options *ro) {
            ctgn_robust_options_default(ro);
            ro->num_iters_icp = options.num_iters_icp;
            ro->min_number_neighbors = options.min_number_neighbors;
            ro->max_number_neighbors = options.max_number_neighbors;
            ro->debug_print = options.debug_print ? 1 : 0;
            ro->max_num_residuals = options.max_num_residuals;
            ro->loss_function = (int32_t) options.loss_function;                             // same enum order, ct_icp.h:41-47
            ro->ls_max_num_iters = options.ls_max_num_iters;
            ro->num_closest_neighbors = options.num_closest_neighbors;
            ro->weight_alpha = options.weight_alpha;
            ro->weight_neighborhood = options.weight_neighborhood;
            ro->power_planarity = options.power_planarity;
            ro->max_dist_to_plane_ct_icp = options.max_dist_to_plane_ct_icp;
            ro->ls_sigma = options.ls_sigma;
            ro->ls_tolerant_min_threshold = options.ls_tolerant_min_threshold;
            ro->threshold_orientation_norm = options.threshold_orientation_norm;
            ro->threshold_translation_norm = options.threshold_translation_norm;
        }

        inline void robust_prior_of(const PreviousFrameMotionModel &model, ctgn_robust_prior *prior) {   // motion_model.cpp:12-61
            const auto &mo = model.GetOptionsConst();
            prior->beta_location_consistency = mo.beta_location_consistency;
            prior->beta_constant_velocity = mo.beta_constant_velocity;
            prior->beta_small_velocity = mo.beta_small_velocity;
            prior->beta_orientation_consistency = mo.beta_orientation_consistency;
            for (int c = 0; c < 3; ++c) {
                prior->previous_begin_tr[c] = model.PreviousFrame().BeginTr()[c];
                prior->previous_end_tr[c] = model.PreviousFrame().EndTr()[c];
            }
            for (int c = 0; c < 4; ++c) prior->previous_end_quat[c] = model.PreviousFrame().EndQuat().coeffs()[c];
        }
    }

    inline std::optional<ICPSummary> GpuGaussNewton(const ISlamMap &voxels_map,
                                                    slam::ProxyView<Eigen::Vector3d> &raw_kpts,
                                                    slam::ProxyView<Eigen::Vector3d> &world_kpts,
                                                    slam::ProxyView<double> &timestamps,
                                                    TrajectoryFrame &frame_to_optimize,
                                                    const AMotionModel *motion_model,
                                                    const CTICPOptions &options) {
        auto *gpu_map = dynamic_cast<const GpuVoxelMap *>(&voxels_map);
        ctgn_view raw, world, ts;
        if (!gpu_map || !ctgn_glue::view_of(raw_kpts, &raw) || !ctgn_glue::view_of(world_kpts, &world) ||
            !ctgn_glue::view_of(timestamps, &ts))
            return std::nullopt;
        ctgn_options co;
        ctgn_glue::gn_options_of(options, &co);
        ctgn_motion_prior prior, *prior_ptr = nullptr;
        if (auto *model = dynamic_cast<const PreviousFrameMotionModel *>(motion_model)) {    // ct_icp.cpp:885-889
            ctgn_glue::gn_prior_of(*model, &prior);
            prior_ptr = &prior;
        }
        double pose[14], tbe[2];
        ctgn_glue::pose_to_array(frame_to_optimize, pose, tbe);
        ctgn_summary s;
        const ctgn_status st = ctgn_register(gpu_map->handle(), raw, const_cast<void *>(world.base), world.stride_bytes, world.dtype, ts,
                                             raw_kpts.size(), pose, tbe, &co, prior_ptr, &s);
        if (st == CTGN_OK) ctgn_glue::array_to_pose(pose, frame_to_optimize);
        return ctgn_glue::to_summary(st, s, gpu_map->handle());
    }

    inline std::optional<ICPSummary> GpuCeres(const ISlamMap &voxels_map,
                                              slam::ProxyView<Eigen::Vector3d> &raw_kpts,
                                              slam::ProxyView<Eigen::Vector3d> &world_kpts,
                                              slam::ProxyView<double> &timestamps,
                                              TrajectoryFrame &frame_to_optimize,
                                              const AMotionModel *previous_frame,
                                              const CTICPOptions &options) {
        auto *gpu_map = dynamic_cast<const GpuVoxelMap *>(&voxels_map);
        ctgn_view raw, world, ts;
        if (!gpu_map || options.parametrization != CONTINUOUS_TIME || options.distance != POINT_TO_PLANE ||
            !ctgn_glue::view_of(raw_kpts, &raw) || !ctgn_glue::view_of(world_kpts, &world) || !ctgn_glue::view_of(timestamps, &ts))
            return std::nullopt;
        ctgn_robust_options ro;
        ctgn_glue::robust_options_of(options, &ro);
        ctgn_robust_prior prior, *prior_ptr = nullptr;
        if (auto *model = dynamic_cast<const PreviousFrameMotionModel *>(previous_frame)) {  // ct_icp.cpp:608-610, motion_model.cpp:12-61
            ctgn_glue::robust_prior_of(*model, &prior);
            prior_ptr = &prior;
        }
        double pose[14], tbe[2];
        ctgn_glue::pose_to_array(frame_to_optimize, pose, tbe);
        ctgn_summary s;
        const ctgn_status st = ctgn_register_robust(gpu_map->handle(), raw, const_cast<void *>(world.base), world.stride_bytes, world.dtype, ts,
                                                    raw_kpts.size(), pose, tbe, &ro, prior_ptr, &s);
        if (st == CTGN_ERR_SOLVER) throw std::runtime_error("Error During Optimization");    // ct_icp.cpp:628-631
        if (st == CTGN_OK) ctgn_glue::array_to_pose(pose, frame_to_optimize);
        return ctgn_glue::to_summary(st, s, gpu_map->handle());
    }

} // namespace ct_icp

#endif //CT_ICP_GN_GPU_ARM_H
