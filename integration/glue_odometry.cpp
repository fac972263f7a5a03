// integration/glue_odometry.cpp — the drop-in one level up from glue_check.cpp: the REFERENCE'S OWN ct_icp::Odometry
// (include/ct_icp/odometry.h:231-272, src/ct_icp/odometry.cpp, compiled where it lies under /root/reference) driving its whole per-frame
// loop — InitializeMotion, InitializeFrame (shuffle + sub_sample_frame), TryRegister (grid_sampling + CT_ICP_Registration::Register),
// AssessRegistration (:604-684), the robust retry loop (:780-852), the full-scan undistortion (:461-486) and UpdateMap (:854-952) —
// on a map that comes out of the reference's own factory call `options.map_options->MakeMapFromOptions()` (odometry.cpp:700):
//   map_kind 0: MultipleResolutionVoxelMap::Options  (`map_type: MULTI_RESOLUTION_VOXEL_HASHMAP`, the reference's CPU map and CPU solver loops)
//   map_kind 1: GpuVoxelMap::Options                 (`map_type: GPU_VOXEL_HASHMAP`, integration/gpu_map.h; Register reaches libctgn.so through
//                                                     the one-line arms of integration/gn_gpu_arm.h, every map call through ISlamMap)
//   map_kind 2: the same with `frame_pipeline` on:   in oracle/_ref/libctgn_ref_odometry_armed.so — odometry.cpp compiled with the four arms of
//                                                     integration/odometry_gpu_arm.h — InitializeFrame, TryRegister, the undistortion loops and
//                                                     the map half of UpdateMap run on the device too (map_kind 1 there: the arms stand down;
//                                                     map_kind 3: kind 2 with `frame_shuffle_on_device`).
//                                                     In the un-armed library the flag has nothing to switch and kind 2 is refused.
// Nothing of Odometry is restated here: the functions below fill an OdometryOptions, construct ct_icp::Odometry and call RegisterFrame.
// extern "C" so that the tests can feed both instances the same scans from Python (tests/test_odometry_glue.py, tests/odometry_vs_reference.py);
// linked into oracle/_ref/libctgn_ref_odometry.so by oracle/Makefile (target `odometry`) with the reference's sources + libctgn.so.
// TEST INFRASTRUCTURE: third-party arithmetic underneath the reference is oracle/shims/ (see oracle/shims/mini_eigen.h).
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <ct_icp/odometry.h>
#include <ct_icp/gpu_map.h>
#ifdef CTGN_GLUE_ARMED
#include <ct_icp/odometry_gpu_arm.h>
#endif

namespace {
    struct OdometryUnderTest {
        ct_icp::OdometryOptions options;
        std::vector<ct_icp::MultipleResolutionVoxelMap::ResolutionParam> resolutions = {{0.8, 0.1, 30}};   // driving_config.yaml:42-46
        double default_radius = 0.75;
        int device = 0;
        int map_kind = 0;
        std::unique_ptr<ct_icp::Odometry> odometry;
        std::ostringstream log;
    };
    thread_local std::string g_error;

    template<typename F> int guarded(F &&f) {
        try { f(); return 0; }
        catch (const ctgn_ref_shim::CheckFailure &e) { g_error = e.what(); return -3; }
        catch (const std::exception &e) { g_error = e.what(); return -4; }
    }

    // one (key, value) at a time: the keys are the reference's own YAML keys (config/odometry/driving_config.yaml, src/ct_icp/config.cpp)
    bool set_option(OdometryUnderTest &u, const std::string &key, double v) {
        auto &o = u.options;
        auto &c = o.ct_icp_options;
        if (key == "voxel_size") o.voxel_size = v;
        else if (key == "sample_voxel_size") o.sample_voxel_size = v;
        else if (key == "init_voxel_size") o.init_voxel_size = v;
        else if (key == "init_sample_voxel_size") o.init_sample_voxel_size = v;
        else if (key == "init_num_frames") o.init_num_frames = (int) v;
        else if (key == "max_num_keypoints") o.max_num_keypoints = (int) v;
        else if (key == "sampling") o.sampling = (ct_icp::sampling::SAMPLING_OPTION) (int) v;
        else if (key == "max_distance") o.max_distance = v;
        else if (key == "distance_error_threshold") o.distance_error_threshold = v;
        else if (key == "orientation_error_threshold") o.orientation_error_threshold = v;
        else if (key == "quit_on_error") o.quit_on_error = v != 0;
        else if (key == "robust_registration") o.robust_registration = v != 0;
        else if (key == "robust_minimal_level") o.robust_minimal_level = (int) v;
        else if (key == "robust_num_attempts") o.robust_num_attempts = (int) v;
        else if (key == "robust_threshold_ego_orientation") o.robust_threshold_ego_orientation = v;
        else if (key == "robust_threshold_relative_orientation") o.robust_threshold_relative_orientation = v;
        else if (key == "robust_relative_trans_threshold") o.robust_relative_trans_threshold = v;
        else if (key == "debug_print") o.debug_print = v != 0;
        else if (key == "with_default_motion_model") o.with_default_motion_model = v != 0;
        else if (key == "beta_location_consistency") o.default_motion_model.beta_location_consistency = v;
        else if (key == "beta_constant_velocity") o.default_motion_model.beta_constant_velocity = v;
        else if (key == "beta_small_velocity") o.default_motion_model.beta_small_velocity = v;
        else if (key == "beta_orientation_consistency") o.default_motion_model.beta_orientation_consistency = v;
        else if (key == "motion_compensation") o.motion_compensation = (ct_icp::MOTION_COMPENSATION) (int) v;
        else if (key == "initialization") o.initialization = (ct_icp::INITIALIZATION) (int) v;
        else if (key == "solver") c.solver = (ct_icp::CT_ICP_SOLVER) (int) v;
        else if (key == "num_iters_icp") c.num_iters_icp = (int) v;
        else if (key == "min_number_neighbors") c.min_number_neighbors = (int) v;
        else if (key == "max_number_neighbors") c.max_number_neighbors = (int) v;
        else if (key == "num_closest_neighbors") c.num_closest_neighbors = (int) v;
        else if (key == "max_num_residuals") c.max_num_residuals = (int) v;
        else if (key == "min_num_residuals") c.min_num_residuals = (int) v;
        else if (key == "max_dist_to_plane_ct_icp") c.max_dist_to_plane_ct_icp = v;
        else if (key == "threshold_orientation_norm") c.threshold_orientation_norm = v;
        else if (key == "threshold_translation_norm") c.threshold_translation_norm = v;
        else if (key == "threshold_voxel_occupancy") c.threshold_voxel_occupancy = (int) v;
        else if (key == "loss_function") c.loss_function = (ct_icp::LEAST_SQUARES) (int) v;
        else if (key == "ls_max_num_iters") c.ls_max_num_iters = (int) v;
        else if (key == "ls_num_threads") c.ls_num_threads = (int) v;
        else if (key == "ls_sigma") c.ls_sigma = v;
        else if (key == "ls_tolerant_min_threshold") c.ls_tolerant_min_threshold = v;
        else if (key == "weight_alpha") c.weight_alpha = v;
        else if (key == "weight_neighborhood") c.weight_neighborhood = v;
        else if (key == "power_planarity") c.power_planarity = v;
        else if (key == "icp_debug_print") c.debug_print = v != 0;
        else if (key == "default_radius") u.default_radius = v;
        else if (key == "device") u.device = (int) v;
        else return false;
        return true;
    }
}

extern "C" {

struct glue_odometry_result {
    double pose[14];                 // optimised begin | end pose: qx qy qz qw tx ty tz each (slam::SE3::Parameters() order)
    double initial_pose[14];         // the estimate InitializeMotion started from
    double relative_distance, relative_orientation, ego_orientation, distance_correction;
    double milliseconds;             // wall time of the RegisterFrame call
    int32_t success, points_added, sample_size, number_of_residuals, number_of_attempts, robust_level, icp_num_iters, num_corrected;
    uint64_t map_points;             // only filled on request (O(map) on the CPU map)
    // where the call's time went, from the reference's own logged_values (odometry.cpp:210-211,428,495-499), milliseconds:
    // [0] odometry_total  [1] odometry_initialization(ms) = InitializeFrame + log  [2] odometry_try_register (0 on the robust-registration
    // path, which does not log it)  [3] odometry_transform(ms) = the undistortion loops  [4] odometry_map_update(ms)  [5] odometry_initialization
    // = compute_frame_info + InitializeMotion
    double phase_ms[6];
    // the arms' own marks (integration/odometry_gpu_arm.h, logged_values odometry_gpu_*; 0 without arms): [0] the shuffles on the host
    // [1] the ctgn_frame_begin call  [2] building the host image of the sampled frame  [3] the whole TryRegister arm  [4] the
    // ctgn_frame_undistort call  [5] the whole undistortion arm (summary vectors + fill loop beside the call)
    double gpu_ms[6];
};

const char *glue_odometry_last_error() { return g_error.c_str(); }

// profile 0: OdometryOptions() with config/odometry/driving_config.yaml's values (the defaults of the struct ARE those values where the
// file does not say otherwise); 1: OdometryOptions::DefaultDrivingProfile(); 2: RobustDrivingProfile(); 3: DefaultRobustOutdoorLowInertia()
void *glue_odometry_options(int profile) {
    auto *u = new OdometryUnderTest();
    if (profile == 1) u->options = ct_icp::OdometryOptions::DefaultDrivingProfile();
    else if (profile == 2) u->options = ct_icp::OdometryOptions::RobustDrivingProfile();
    else if (profile == 3) u->options = ct_icp::OdometryOptions::DefaultRobustOutdoorLowInertia();
    if (profile == 0) {                                                    // driving_config.yaml:18-90
        auto &c = u->options.ct_icp_options;
        c.solver = ct_icp::CERES;
        c.num_iters_icp = 5; c.max_num_residuals = 900; c.min_num_residuals = 100; c.weight_alpha = 0.9; c.weight_neighborhood = 0.1;
        c.min_number_neighbors = 20; c.max_number_neighbors = 20; c.num_closest_neighbors = 1; c.power_planarity = 2;
        c.threshold_voxel_occupancy = 1; c.threshold_orientation_norm = 0.1; c.threshold_translation_norm = 0.01;
        c.loss_function = ct_icp::CAUCHY; c.ls_max_num_iters = 5; c.ls_num_threads = 6; c.ls_sigma = 0.1; c.ls_tolerant_min_threshold = 0.05;
        c.debug_print = false;
    }
    u->options.debug_print = false;
    u->options.ct_icp_options.debug_print = false;
    return u;
}

int glue_odometry_set(void *h, const char *key, double value) {
    auto *u = static_cast<OdometryUnderTest *>(h);
    if (!set_option(*u, key, value)) { g_error = std::string("unknown option ") + key; return -1; }
    return 0;
}

int glue_odometry_set_resolutions(void *h, int n, const double *resolution, const double *min_distance, const int32_t *max_num_points) {
    auto *u = static_cast<OdometryUnderTest *>(h);
    u->resolutions.clear();
    for (int i = 0; i < n; ++i) u->resolutions.push_back({resolution[i], min_distance[i], max_num_points[i]});
    return 0;
}

// The map options object decides which map Odometry's constructor builds (odometry.cpp:699-700); nothing else differs between the two kinds.
int glue_odometry_start(void *h, int map_kind) {
    auto *u = static_cast<OdometryUnderTest *>(h);
    return guarded([&] {
        if (map_kind == 0) {
            auto mo = std::make_shared<ct_icp::MultipleResolutionVoxelMap::Options>();
            mo->resolutions = u->resolutions;
            mo->default_radius = u->default_radius;
            mo->max_frames_to_keep = 1;                                    // src/ct_icp/map.cpp:63 reads it; the old loader sets 1 (:18)
            u->options.map_options = mo;
        } else {
#ifndef CTGN_GLUE_ARMED
            if (map_kind >= 2) throw std::runtime_error("map kinds 2 and 3 need the armed library (oracle/_ref/libctgn_ref_odometry_armed.so)");
#endif
            auto mo = std::make_shared<ct_icp::GpuVoxelMap::Options>();
            mo->resolutions = u->resolutions;
            mo->default_radius = u->default_radius;
            mo->device = u->device;
            mo->frame_pipeline = map_kind >= 2;
            mo->frame_shuffle_on_device = map_kind == 3;                   // kind 3: kind 2 with the first shuffle made on the device
            u->options.map_options = mo;
        }
        u->map_kind = map_kind;
        u->odometry = std::make_unique<ct_icp::Odometry>(u->options);
        const std::string want = map_kind == 0 ? "MULTI_RESOLUTION_VOXEL_HASHMAP" : "GPU_VOXEL_HASHMAP";
        if (u->options.map_options->GetType() != want) throw std::runtime_error("map options of the wrong type");
        const bool is_gpu = dynamic_cast<ct_icp::GpuVoxelMap *>(u->odometry->GetMapPointer().get()) != nullptr;
        if (is_gpu != (map_kind >= 1)) throw std::runtime_error("Odometry built the other map kind");
    });
}

// 1 in oracle/_ref/libctgn_ref_odometry_armed.so, 0 in the un-armed library
int glue_odometry_is_armed() {
#ifdef CTGN_GLUE_ARMED
    return 1;
#else
    return 0;
#endif
}

void glue_odometry_destroy(void *h) { delete static_cast<OdometryUnderTest *>(h); }

// One Odometry::RegisterFrame(const slam::PointCloud&, frame_id) call (odometry.cpp:209-224). xyz: n x 3 raw points, t: n timestamps.
// world_out (optional): n x 3, summary.all_corrected_points' world points (the undistorted scan, odometry.cpp:461-476).
// sampled_raw_out (optional, capacity n x 3): the raw points of summary.corrected_points, i.e. the sampled frame (out->num_corrected rows).
int glue_odometry_register_frame(void *h, const double *xyz, const double *t, size_t n, int frame_id, glue_odometry_result *out, double *world_out,
                                 int want_map_points, double *sampled_raw_out) {
    auto *u = static_cast<OdometryUnderTest *>(h);
    return guarded([&] {
        if (!u->odometry) throw std::runtime_error("glue_odometry_start was not called");
        // a frame as the reference's dataset readers hand it over: double xyz + a timestamps field (the vector<WPoint3D> overload of
        // RegisterFrame, odometry.cpp:259-273, wraps the vector without registering its timestamps and trips pointcloud.h:231)
        auto frame = slam::PointCloud::DefaultXYZPtr<double>();
        frame->resize(n);
        frame->AddDefaultTimestampsField();
        {
            auto xyz_view = frame->XYZ<double>();
            auto t_view = frame->TimestampsProxy<double>();
            for (size_t i = 0; i < n; ++i) {
                xyz_view[i] = Eigen::Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
                t_view[i] = t[i];
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        auto summary = u->odometry->RegisterFrame(*frame, (slam::frame_id_t) frame_id);
        out->milliseconds = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < 7; ++i) {
            out->pose[i] = summary.frame.begin_pose.pose[i];
            out->pose[7 + i] = summary.frame.end_pose.pose[i];
            out->initial_pose[i] = summary.initial_frame.begin_pose.pose[i];
            out->initial_pose[7 + i] = summary.initial_frame.end_pose.pose[i];
        }
        out->relative_distance = summary.relative_distance;
        out->relative_orientation = summary.relative_orientation;
        out->ego_orientation = summary.ego_orientation;
        out->distance_correction = summary.distance_correction;
        out->success = summary.success ? 1 : 0;
        out->points_added = summary.points_added ? 1 : 0;
        out->sample_size = summary.sample_size;
        out->number_of_residuals = summary.number_of_residuals;
        out->number_of_attempts = summary.number_of_attempts;
        out->robust_level = summary.robust_level;
        out->icp_num_iters = summary.icp_summary.num_iters;
        out->num_corrected = (int32_t) summary.corrected_points.size();
        out->map_points = want_map_points ? (uint64_t) u->odometry->MapSize() : 0;
        const char *keys[6] = {"odometry_total", "odometry_initialization(ms)", "odometry_try_register", "odometry_transform(ms)",
                               "odometry_map_update(ms)", "odometry_initialization"};
        for (int k = 0; k < 6; ++k) {
            auto it = summary.logged_values.find(keys[k]);
            out->phase_ms[k] = it == summary.logged_values.end() ? 0.0 : it->second;
        }
        const char *gpu_keys[6] = {"odometry_gpu_shuffle", "odometry_gpu_frame_begin", "odometry_gpu_build_frame", "odometry_gpu_try_register",
                                   "odometry_gpu_undistort_call", "odometry_gpu_undistort_arm"};
        for (int k = 0; k < 6; ++k) {
            auto it = summary.logged_values.find(gpu_keys[k]);
            out->gpu_ms[k] = it == summary.logged_values.end() ? 0.0 : it->second;
        }
        if (sampled_raw_out)
            for (size_t i = 0; i < summary.corrected_points.size() && i < n; ++i)
                for (int c = 0; c < 3; ++c) sampled_raw_out[3 * i + c] = summary.corrected_points[i].RawPoint()[c];
        if (world_out && summary.all_corrected_points.size() == n)
            for (size_t i = 0; i < n; ++i)
                for (int c = 0; c < 3; ++c) world_out[3 * i + c] = summary.all_corrected_points[i].WorldPoint()[c];
    });
}

// The local map as a point set (ISlamMap::MapAsPointCloud through Odometry::GetMapPointCloud, odometry.cpp:692-694); call with out = NULL for the size.
int glue_odometry_map_points(void *h, double *out, uint64_t capacity, uint64_t *n_out) {
    auto *u = static_cast<OdometryUnderTest *>(h);
    return guarded([&] {
        auto pc = u->odometry->GetMapPointCloud();
        *n_out = (uint64_t) pc->size();
        if (!out) return;
        auto xyz = pc->WorldPointsProxy<Eigen::Vector3d>();
        for (size_t i = 0; i < pc->size() && i < capacity; ++i) {
            Eigen::Vector3d p = xyz[i];
            out[3 * i] = p[0]; out[3 * i + 1] = p[1]; out[3 * i + 2] = p[2];
        }
    });
}

}
