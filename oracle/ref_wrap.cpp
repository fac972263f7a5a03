// oracle/_ref : extern "C" entry points around the REFERENCE'S OWN CLASSES, compiled from the sources where they lie
// under /root/reference (oracle/Makefile target `_ref`) against the third-party shims of oracle/shims/.
//
// TEST INFRASTRUCTURE ONLY.  Nothing here restates the reference: every function below constructs the reference's
// objects and calls the reference's methods --
//   ct_icp::MultipleResolutionVoxelMap           include/ct_icp/map.h:100-607   (InsertPointCloud / InsertPointInVoxelMap,
//                                                RemoveElementsFarFromLocation, RadiusSearch, GetMapPoints)
//   ct_icp::CT_ICP_Registration::Register        src/ct_icp/ct_icp.cpp:1026-1038 -> DoRegisterGaussNewton :709-996,
//                                                DoRegisterCeres :457-707
//   ct_icp::PreviousFrameMotionModel             include/ct_icp/motion_model.h:33-84, src/ct_icp/motion_model.cpp:12-117
//   slam::Neighborhood::ComputeNeighborhood      include/SlamCore/experimental/neighborhood.h:225-316
//   ct_icp::sub_sample_frame                     src/ct_icp/ct_icp.cpp:65-83
//   slam::Pose::InterpolatePose / operator*      include/SlamCore/types.h:353-366,453-470
// What is NOT the reference in the resulting binary is the arithmetic of its third-party libraries (Eigen, Ceres,
// tsl::robin_map, glog), which oracle/shims/ restates -- see the header of oracle/shims/mini_eigen.h.
// glog's CHECK aborts the process in the reference; the shim throws and the wrappers below return -3 with the message.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <ct_icp/ct_icp.h>
#include <ct_icp/map.h>
#include <ct_icp/motion_model.h>
#include <SlamCore/experimental/neighborhood.h>
#include <SlamCore/pointcloud.h>

namespace {
    struct RefMap {
        ct_icp::MultipleResolutionVoxelMap::Options options;
        std::shared_ptr<ct_icp::MultipleResolutionVoxelMap> map;
        size_t frame_counter = 0;
    };
    thread_local std::string g_last_error;

    template<typename F> int guarded(F &&f) {
        try { f(); return 0; }
        catch (const ctgn_ref_shim::CheckFailure &e) { g_last_error = e.what(); return -3; }
        catch (const std::exception &e) { g_last_error = e.what(); return -4; }
    }

    void set_frame(ct_icp::TrajectoryFrame &frame, const double pose[14], const double t_begin_end[2]) {
        // Eigen::Quaterniond(w, x, y, z); pose14 = (qx qy qz qw tx ty tz) x {begin, end}
        frame.begin_pose = slam::Pose(slam::SE3(Eigen::Quaterniond(pose[3], pose[0], pose[1], pose[2]),
                                                Eigen::Vector3d(pose[4], pose[5], pose[6])), t_begin_end[0], 0);
        frame.end_pose = slam::Pose(slam::SE3(Eigen::Quaterniond(pose[10], pose[7], pose[8], pose[9]),
                                              Eigen::Vector3d(pose[11], pose[12], pose[13])), t_begin_end[1], 0);
        // slam::SE3's constructor normalises the quaternion (types.h:108-110); the callers hand in unit quaternions,
        // and DoRegisterGaussNewton normalises again at ct_icp.cpp:716-717.
    }

    void get_frame(const ct_icp::TrajectoryFrame &frame, double pose[14]) {
        const auto &qb = frame.begin_pose.pose.quat, &qe = frame.end_pose.pose.quat;
        const auto &tb = frame.begin_pose.pose.tr, &te = frame.end_pose.pose.tr;
        pose[0] = qb.x(); pose[1] = qb.y(); pose[2] = qb.z(); pose[3] = qb.w();
        pose[4] = tb.x(); pose[5] = tb.y(); pose[6] = tb.z();
        pose[7] = qe.x(); pose[8] = qe.y(); pose[9] = qe.z(); pose[10] = qe.w();
        pose[11] = te.x(); pose[12] = te.y(); pose[13] = te.z();
    }
}

extern "C" {

const char *ref_last_error() { return g_last_error.c_str(); }

// sizeof / offsets of the reference's PODs as compiled here (the ABI tests compare them with include/ctgn.h's mirrors)
void ref_layout(size_t out[16]) {
    out[0] = sizeof(slam::WPoint3D);
    out[1] = offsetof(slam::WPoint3D, raw_point.point);
    out[2] = offsetof(slam::WPoint3D, raw_point.timestamp);
    out[3] = offsetof(slam::WPoint3D, world_point);
    out[4] = offsetof(slam::WPoint3D, index_frame);
    out[5] = sizeof(slam::Pose);
    out[6] = offsetof(slam::Pose, pose);
    out[7] = offsetof(slam::Pose, ref_timestamp);
    out[8] = offsetof(slam::Pose, dest_timestamp);
    out[9] = offsetof(slam::Pose, ref_frame_id);
    out[10] = offsetof(slam::Pose, dest_frame_id);
    out[11] = sizeof(ct_icp::TrajectoryFrame);
    out[12] = sizeof(slam::SE3);
    out[13] = offsetof(slam::SE3, quat);
    out[14] = offsetof(slam::SE3, tr);
    out[15] = sizeof(slam::Voxel);
}

void *ref_map_create(int n_res, const double *resolution, const double *min_dist, const int *max_points, double default_radius) {
    auto *m = new RefMap();
    m->options.resolutions.resize(size_t(n_res));
    for (int i = 0; i < n_res; ++i)
        m->options.resolutions[size_t(i)] = ct_icp::MultipleResolutionVoxelMap::ResolutionParam{resolution[i], min_dist[i], max_points[i]};
    m->options.default_radius = default_radius;
    m->options.max_frames_to_keep = 1;
    // the reference's own factory: IMapOptions::MakeMapFromOptions (map.h:127-133)
    m->map = std::dynamic_pointer_cast<ct_icp::MultipleResolutionVoxelMap>(m->options.MakeMapFromOptions());
    return m;
}

void ref_map_destroy(void *h) { delete static_cast<RefMap *>(h); }

// mode 0: the per-point body of InsertPointCloud's loop (map.h:207-218): InsertPointInVoxelMap for every resolution;
//         out_inserted[n_res * i + r] = 1 iff the point went into level r.
// mode 1: the whole ISlamMap::InsertPointCloud(pointcloud, out_indices) entry point (map.h:299-302 -> :153-254) on a
//         slam::PointCloud whose world-point field is the xyz buffer (also runs the per-voxel normal estimation).
int ref_map_insert(void *h, const double *xyz, size_t n, int mode, unsigned char *out_inserted) {
    auto *m = static_cast<RefMap *>(h);
    return guarded([&] {
        const size_t n_res = m->options.resolutions.size();
        if (mode == 0) {
            const size_t fidx = m->frame_counter++;
            for (size_t i = 0; i < n; ++i) {
                Eigen::Vector3d p(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
                for (size_t r = 0; r < n_res; ++r) {
                    auto voxel = m->map->InsertPointInVoxelMap(p, r, fidx, i);
                    if (out_inserted) out_inserted[n_res * i + r] = voxel.has_value() ? 1 : 0;
                }
            }
        } else {
            auto pc = slam::PointCloud::DefaultXYZPtr<double>();
            pc->resize(n);
            pc->SetWorldPointsField(slam::PointCloud::Field{pc->GetXYZField()});
            auto view = pc->XYZ<double>();
            for (size_t i = 0; i < n; ++i) view[i] = Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
            std::vector<size_t> out_indices;
            m->map->InsertPointCloud(*pc, out_indices);
            m->frame_counter++;
        }
    });
}

int ref_map_remove_far(void *h, const double location[3], double distance) {
    auto *m = static_cast<RefMap *>(h);
    return guarded([&] { m->map->RemoveElementsFarFromLocation(Eigen::Vector3d(location[0], location[1], location[2]), distance); });
}

void ref_map_clear(void *h) { static_cast<RefMap *>(h)->map->ClearMap(); }

size_t ref_map_num_points(void *h, int res_index) {
    auto *m = static_cast<RefMap *>(h);
    if (res_index == 0) return m->map->NumPoints();
    size_t n = 0;
    guarded([&] { n = m->map->GetMapPoints(size_t(res_index))->size(); });
    return n;
}

// GetMapPoints(res_index) (map.h:356-380); the ORDER is the hash table's iteration order (shimmed container) -> compare as sets
int ref_map_export(void *h, int res_index, double *out_xyz, size_t capacity, size_t *out_n) {
    auto *m = static_cast<RefMap *>(h);
    return guarded([&] {
        auto pc = m->map->GetMapPoints(size_t(res_index));
        auto view = pc->XYZ<double>();
        *out_n = pc->size();
        for (size_t i = 0; i < pc->size() && i < capacity; ++i) {
            Eigen::Vector3d p = view[i];
            out_xyz[3 * i] = p.x(); out_xyz[3 * i + 1] = p.y(); out_xyz[3 * i + 2] = p.z();
        }
    });
}

void ref_map_search_params(void *h, double radius, double out[4]) {
    auto *m = static_cast<RefMap *>(h);
    auto params = m->map->SearchParamsFromRadiusSearch(radius);
    out[0] = params.radius; out[1] = params.voxel_resolution; out[2] = double(params.map_id); out[3] = double(params.voxel_neighborhood);
}

// radius <= 0: ISlamMap::ComputeNeighborhood(query, k) (default radius, the call of ct_icp.cpp:763);
// radius  > 0: ISlamMap::RadiusSearch(query, radius, k) (map.h:516-522).  Points come back in the reference's order
// (farthest first).  out_xyz[(i * k + j) * 3 ..], out_count[i].
int ref_radius_search(void *h, const double *queries, size_t n, double radius, int k, int *out_count, double *out_xyz) {
    auto *m = static_cast<RefMap *>(h);
    return guarded([&] {
        const ct_icp::ISlamMap &map = *m->map;
        for (size_t i = 0; i < n; ++i) {
            Eigen::Vector3d q(queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]);
            slam::Neighborhood nb = radius > 0. ? map.RadiusSearch(q, radius, k, true, nullptr) : map.ComputeNeighborhood(q, k);
            out_count[i] = int(nb.points.size());
            for (size_t j = 0; j < nb.points.size(); ++j)
                for (int c = 0; c < 3; ++c) out_xyz[(i * size_t(k) + j) * 3 + size_t(c)] = nb.points[j][c];
        }
    });
}

// slam::Neighborhood::ComputeNeighborhood(A2D | NORMAL) on explicit points (neighborhood.h:225-316); returns is_valid
int ref_neighborhood(const double *points, size_t n, double out_normal[3], double *out_a2d) {
    slam::Neighborhood nb;
    nb.points.resize(n);
    for (size_t i = 0; i < n; ++i) nb.points[i] = Eigen::Vector3d(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
    nb.ComputeNeighborhood(slam::A2D | slam::NORMAL);
    if (!nb.is_valid) return 0;
    for (int c = 0; c < 3; ++c) out_normal[c] = nb.description.normal[c];
    *out_a2d = nb.description.a2D;
    return 1;
}

// slam::Voxel::Coordinates (types.cxx:13-20) and std::hash<slam::Voxel> (types.h:610-623)
void ref_voxel_coordinates(const double p[3], double voxel_size, int out[3], unsigned long long *out_hash) {
    slam::Voxel v = slam::Voxel::Coordinates(Eigen::Vector3d(p[0], p[1], p[2]), voxel_size);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
    *out_hash = (unsigned long long) std::hash<slam::Voxel>()(v);
}

// TPose::GetAlphaTimestamp (types.h:192-219)
double ref_alpha_timestamp(double t, double t_begin, double t_end) {
    slam::Pose b, e;
    b.dest_timestamp = t_begin; e.dest_timestamp = t_end;
    return b.GetAlphaTimestamp(t, e);
}

// world_i = begin.InterpolatePose(end, t_i) * raw_i  (the statement of ct_icp.cpp:964-966 / odometry.cpp:461-486)
int ref_transform_points(const double pose[14], const double t_begin_end[2], const double *t, const double *raw, size_t n, double *out_world) {
    return guarded([&] {
        ct_icp::TrajectoryFrame frame;
        set_frame(frame, pose, t_begin_end);
        for (size_t i = 0; i < n; ++i) {
            Eigen::Vector3d w = frame.begin_pose.InterpolatePose(frame.end_pose, t[i]) * Eigen::Vector3d(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]);
            out_world[3 * i] = w.x(); out_world[3 * i + 1] = w.y(); out_world[3 * i + 2] = w.z();
        }
    });
}

// ct_icp::sub_sample_frame (ct_icp.cpp:65-83).  The surviving points are identified through index_frame (set to the input
// index here); the output ORDER is the shimmed hash table's -> compare as sets.
int ref_sub_sample_frame(const double *raw, size_t n, double voxel_size, unsigned int *out_indices, size_t *out_n) {
    return guarded([&] {
        std::vector<slam::WPoint3D> frame(n);
        for (size_t i = 0; i < n; ++i) {
            frame[i].RawPoint() = Eigen::Vector3d(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]);
            frame[i].index_frame = (unsigned int) i;
        }
        ct_icp::sub_sample_frame(frame, voxel_size);
        *out_n = frame.size();
        for (size_t i = 0; i < frame.size(); ++i) out_indices[i] = frame[i].index_frame;
    });
}

struct ref_options {                // the CTICPOptions fields the two routes read
    int solver;                     // 0 GN, 1 CERES  (ct_icp::CT_ICP_SOLVER)
    int num_iters_icp, min_number_neighbors, max_number_neighbors;
    double max_dist_to_plane_ct_icp, threshold_orientation_norm, threshold_translation_norm;
    // CERES route
    int loss_function;              // ct_icp::LEAST_SQUARES
    int ls_max_num_iters, ls_num_threads, max_num_residuals, num_closest_neighbors;
    double ls_sigma, ls_tolerant_min_threshold, weight_alpha, weight_neighborhood, power_planarity;
    int point_to_plane_with_distortion;
};
struct ref_prior {                  // PreviousFrameMotionModel::Options + the previous frame's poses
    double beta_location_consistency, beta_constant_velocity, beta_small_velocity, beta_orientation_consistency;
    double previous_pose[14];
    double previous_t_begin_end[2];
};
struct ref_summary {
    int success, num_residuals_used, num_iters;
    char error_log[256];
};

// CT_ICP_Registration::Register(map, std::vector<slam::WPoint3D>&, TrajectoryFrame&, motion_model) -- the vector overload
// Odometry::TryRegister uses (odometry.cpp:573-579).  wpoints: n records of sizeof(slam::WPoint3D) bytes, read AND
// written (world_point is rewritten by the solver).
int ref_register(void *h, void *wpoints, size_t n, double pose_io[14], const double t_begin_end[2], const ref_options *o,
                 const ref_prior *prior, ref_summary *out) {
    auto *m = static_cast<RefMap *>(h);
    std::memset(out, 0, sizeof(*out));
    return guarded([&] {
        std::vector<slam::WPoint3D> keypoints(n);
        std::memcpy(static_cast<void *>(keypoints.data()), wpoints, n * sizeof(slam::WPoint3D));
        ct_icp::TrajectoryFrame frame;
        set_frame(frame, pose_io, t_begin_end);

        ct_icp::CT_ICP_Registration registration;
        auto &opt = registration.Options();
        opt.solver = o->solver == 0 ? ct_icp::GN : ct_icp::CERES;
        opt.num_iters_icp = o->num_iters_icp;
        opt.min_number_neighbors = o->min_number_neighbors;
        opt.max_number_neighbors = o->max_number_neighbors;
        opt.max_dist_to_plane_ct_icp = o->max_dist_to_plane_ct_icp;
        opt.threshold_orientation_norm = o->threshold_orientation_norm;
        opt.threshold_translation_norm = o->threshold_translation_norm;
        opt.loss_function = ct_icp::LEAST_SQUARES(o->loss_function);
        opt.ls_max_num_iters = o->ls_max_num_iters;
        opt.ls_num_threads = o->ls_num_threads > 0 ? o->ls_num_threads : 1;
        opt.max_num_residuals = o->max_num_residuals;
        opt.num_closest_neighbors = o->num_closest_neighbors;
        opt.ls_sigma = o->ls_sigma;
        opt.ls_tolerant_min_threshold = o->ls_tolerant_min_threshold;
        opt.weight_alpha = o->weight_alpha;
        opt.weight_neighborhood = o->weight_neighborhood;
        opt.power_planarity = o->power_planarity;
        opt.point_to_plane_with_distortion = o->point_to_plane_with_distortion != 0;
        opt.parametrization = ct_icp::CONTINUOUS_TIME;
        opt.distance = ct_icp::POINT_TO_PLANE;
        opt.debug_print = false;

        ct_icp::PreviousFrameMotionModel model;
        const ct_icp::AMotionModel *model_ptr = nullptr;
        if (prior) {
            auto &mo = model.GetOptions();
            mo.beta_location_consistency = prior->beta_location_consistency;
            mo.beta_constant_velocity = prior->beta_constant_velocity;
            mo.beta_small_velocity = prior->beta_small_velocity;
            mo.beta_orientation_consistency = prior->beta_orientation_consistency;
            ct_icp::TrajectoryFrame previous;
            set_frame(previous, prior->previous_pose, prior->previous_t_begin_end);
            model.UpdateState(previous, 0);
            model_ptr = &model;
        }

        ct_icp::ICPSummary summary = registration.Register(*m->map, keypoints, frame, model_ptr, nullptr);
        out->success = summary.success ? 1 : 0;
        out->num_residuals_used = summary.num_residuals_used;
        out->num_iters = summary.num_iters;
        std::strncpy(out->error_log, summary.error_log.c_str(), sizeof(out->error_log) - 1);
        get_frame(frame, pose_io);
        std::memcpy(wpoints, static_cast<const void *>(keypoints.data()), n * sizeof(slam::WPoint3D));
    });
}

}  // extern "C"
