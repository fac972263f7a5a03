// integration/odometry_gpu_arm.h — the GPU arms of ct_icp::Odometry::DoRegister (src/ct_icp/odometry.cpp:386-501): the scan-sized loops either
// side of the registration handed to the same libctgn.so handle the map lives on, with the scan resident on the device from
// InitializeFrame to UpdateMap. A maintainer drops this file into the reference tree as include/ct_icp/odometry_gpu_arm.h and adds
// FOUR statements to src/ct_icp/odometry.cpp (after `#include <ct_icp/odometry_gpu_arm.h>`):
//
//   std::vector<slam::WPoint3D> Odometry::InitializeFrame(...) {                                          // :333, first statement
//       if (auto gpu_frame = GpuInitializeFrame(map_.get(), const_frame, frame_info, options_, trajectory_[frame_info.registered_fid], g_))
//           return std::move(*gpu_frame);
//   void Odometry::TryRegister(...) {                                                                     // :525, first statement
//       if (GpuTryRegister(map_.get(), frame, frame_info, options, registration_summary, sample_voxel_size, motion_model, options_, g_,
//                          !callbacks_.empty())) return;
//   // Distort the Frame using the current estimate                                                       // :461, the two loops become the else branch
//       if (!GpuUndistortFrame(map_.get(), const_frame, frame_info, frame, summary, options_.ct_icp_options.ls_num_threads)) { ... :462-486 ... }
//   map_->RemoveElementsFarFromLocation(location, kMaxDistance);                                          // :940, in front of it
//       if (GpuUpdateMap(map_.get(), location, kMaxDistance, add_points)) {
//           if (add_points) insertion_tracker_.InsertFrame(registered_fid); else insertion_tracker_.SkipFrame();
//           return;
//       }
//
// and, optionally, a FIFTH in the file-scope lambda compute_frame_info (:186-197), in front of its std::minmax_element:
//       if (GpuFrameTimeRange(timestamps, frame_info.begin_timestamp, frame_info.end_timestamp)) return frame_info;
//   (the same minimum and maximum read through the strided view the arms upload from: 0.03 ms for a 133 k-point scan instead of the 0.42 ms
//    the proxy iterators take — a third of what is left of an armed frame. No device work: it only spares the host a type switch per element.)
//
// Everything else of Odometry is untouched and keeps deciding: InitializeMotion, the startup regimen, AssessRegistration (:604-684), the
// robust retry loop (:780-852, it calls TryRegister — i.e. the arm — once per attempt on the same resident sampled frame), the insertion
// policy of UpdateMap (:855-934), the trajectory. Every arm returns "not mine" (std::nullopt / false) and the reference's own code runs
// when the map is not a GpuVoxelMap with `frame_pipeline`, when motion_compensation is not CONTINUOUS, for `sampling: ADAPTIVE`, for the
// ROBUST solver or a CERES configuration libctgn has no route for, or when callbacks are registered (they are handed keypoints before
// the registration, which the device only reports after it). oracle/Makefile applies exactly these insertions to the reference's
// odometry.cpp on its way into the compiler (GLUE_PATCH2) and links the result into oracle/_ref/libctgn_ref_odometry_armed.so;
// tests/test_odometry_glue.py runs it beside the un-armed library.
//
// The random stream. Odometry shuffles with its own std::mt19937_64 g_: the frame before sub_sample_frame (:349 — which point of a voxel
// survives), the sampled frame (:361 — only the order robin_map's iteration left) and, above max_num_keypoints, the keypoints (:550).
// std::shuffle's draws depend on the length of the range alone, so shuffling an INDEX vector with g_ gives the permutation the
// reference applies to the points: the first shuffle is reproduced exactly and the sampled-frame SET of every frame is the reference's.
// The other two shuffle what is already a uniformly random order here (the sampled frame stays in processing order, the keypoints are
// its first points per voxel): the arms draw the same numbers from g_ (so the stream stays in step with an un-armed run, frame after
// frame) and keep their order. std::shuffle of 132 k indices costs about a millisecond of one host core — as much as everything the
// device does for the frame. `frame_shuffle_on_device` (GpuVoxelMap::Options, off by default) trades the reference's permutation for one
// of the same kind made on the GPU (ctgn_frame_options::shuffle_seed, seeded with one draw from g_): a different, equally random choice
// of the surviving points, and no host shuffle at all.
#ifndef CT_ICP_ODOMETRY_GPU_ARM_H
#define CT_ICP_ODOMETRY_GPU_ARM_H

#include <algorithm>
#include <chrono>
#include <numeric>
#include <optional>
#include <random>

#include <ct_icp/odometry.h>
#include <ct_icp/gn_gpu_arm.h>

namespace ct_icp {

    namespace ctgn_glue {
        inline GpuVoxelMap *frame_pipeline_of(ISlamMap *map, const OdometryOptions &options) {
            auto *gpu_map = dynamic_cast<GpuVoxelMap *>(map);
            if (!gpu_map || !gpu_map->GetOptions().frame_pipeline || !gpu_map->GetOptions().device_updates) return nullptr;
            if (options.motion_compensation != CONTINUOUS) return nullptr;       // TransformPoint's other modes (odometry.cpp:171-184)
            if (options.sampling == sampling::ADAPTIVE) return nullptr;
            const auto &icp = options.ct_icp_options;
            if (icp.solver == ROBUST) return nullptr;
            if (icp.solver == CERES && (icp.parametrization != CONTINUOUS_TIME || icp.distance != POINT_TO_PLANE)) return nullptr;
            return gpu_map;
        }

        inline void fatal_unless_ok(ctgn_status st, ctgn_handle h) {             // the reference CHECK-aborts where libctgn reports
            SLAM_CHECK_STREAM(st == CTGN_OK, "libctgn: " << ctgn_last_error(h));
        }

        // element i of a view whose source is FLOAT32 / FLOAT64 (what view_of accepted), without the per-access dispatch of a ProxyView
        inline Eigen::Vector3d point_of(const ctgn_view &v, size_t i) {
            const char *p = static_cast<const char *>(v.base) + i * v.stride_bytes;
            if (v.dtype == CTGN_F64) { const double *q = reinterpret_cast<const double *>(p); return Eigen::Vector3d(q[0], q[1], q[2]); }
            const float *q = reinterpret_cast<const float *>(p);
            return Eigen::Vector3d(q[0], q[1], q[2]);
        }

        inline double scalar_of(const ctgn_view &v, size_t i) {
            const char *p = static_cast<const char *>(v.base) + i * v.stride_bytes;
            return v.dtype == CTGN_F64 ? *reinterpret_cast<const double *>(p) : (double) *reinterpret_cast<const float *>(p);
        }

        // smallest and largest element of a FLOAT32 / FLOAT64 view (std::minmax_element's VALUES: which of several equal elements it points
        // at does not matter to compute_frame_info, which dereferences both at once)
        inline void minmax_of(const ctgn_view &v, size_t n, double &lo, double &hi) {
            lo = hi = scalar_of(v, 0);
            if (v.dtype == CTGN_F64) {
                const char *p = static_cast<const char *>(v.base);
                for (size_t i = 1; i < n; ++i) {
                    const double x = *reinterpret_cast<const double *>(p + i * v.stride_bytes);
                    lo = x < lo ? x : lo;
                    hi = x < hi ? hi : x;          // (minmax_element keeps the LAST of equal maxima: `!(x < hi)` moves on)
                }
            } else {
                for (size_t i = 1; i < n; ++i) {
                    const double x = scalar_of(v, i);
                    lo = x < lo ? x : lo;
                    hi = x < hi ? hi : x;
                }
            }
        }

        inline double ms_since(std::chrono::steady_clock::time_point t0) {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }

        // std::shuffle(first, last, g) for a range whose CONTENT does not matter: the same draws from g
        inline void advance_like_shuffle(std::vector<uint32_t> &scratch, size_t n, std::mt19937_64 &g) {
            scratch.resize(n);
            std::shuffle(scratch.begin(), scratch.end(), g);
        }

        // InitializeFrame's transform (odometry.cpp:371-375) for the host image of the resident sampled frame, on demand: the world points
        // of `frame` under `estimate`, from the device
        inline void fetch_initial_world_points(GpuVoxelMap &gpu_map, std::vector<slam::WPoint3D> &frame, const TrajectoryFrame &estimate) {
            auto &session = gpu_map.frame_session();
            if (session.world_initialised || frame.empty()) return;
            double pose[14], tbe[2];
            pose_to_array(estimate, pose, tbe);
            ctgn_frame_outputs out{};
            out.sampled_world_base = frame[0].world_point.data();
            out.sampled_world_stride_bytes = sizeof(slam::WPoint3D);
            out.sampled_world_dtype = CTGN_F64;
            fatal_unless_ok(ctgn_frame_undistort(gpu_map.handle(), pose, tbe, &out), gpu_map.handle());
            session.world_initialised = true;
        }
    }

    // compute_frame_info (odometry.cpp:186-197), optional fifth statement: first and last timestamp of the scan without the proxy iterators.
    // "Not mine" when no GpuVoxelMap with `frame_pipeline` is alive in the process (the lambda has no Odometry to ask) or when the
    // timestamps are not FLOAT32 / FLOAT64 — the reference's std::minmax_element then runs as before.
    template<typename TimestampsView>
    inline bool GpuFrameTimeRange(const TimestampsView &timestamps, double &begin_timestamp, double &end_timestamp) {
        if (ctgn_glue::frame_pipeline_maps().load(std::memory_order_relaxed) <= 0 || timestamps.empty()) return false;
        ctgn_view ts;
        if (!ctgn_glue::view_of(timestamps, &ts)) return false;
        ctgn_glue::minmax_of(ts, timestamps.size(), begin_timestamp, end_timestamp);
        return true;
    }

    // InitializeFrame (odometry.cpp:333-382): shuffle, sub_sample_frame, timestamps of the first two frames, initial transform.
    inline std::optional<std::vector<slam::WPoint3D>> GpuInitializeFrame(ISlamMap *map, const slam::PointCloud &const_frame,
                                                                         const Odometry::FrameInfo &frame_info, const OdometryOptions &options,
                                                                         const TrajectoryFrame &tr_frame, std::mt19937_64 &g) {
        auto *gpu_map = ctgn_glue::frame_pipeline_of(map, options);
        if (!gpu_map) return std::nullopt;
        auto &session = gpu_map->frame_session();
        session.active = false;
        session.undistorted = false;
        const auto view_timestamps = const_frame.TimestampsProxy<double>();
        const auto view_xyz = const_frame.XYZConst<double>();
        ctgn_view raw, ts;
        const size_t n = const_frame.size();
        if (n == 0 || !ctgn_glue::view_of(view_xyz, &raw) || !ctgn_glue::view_of(view_timestamps, &ts)) return std::nullopt;

        const auto kIndexFrame = frame_info.registered_fid;
        const bool kIsAtStartup = kIndexFrame < options.init_num_frames;
        const bool shuffle_on_device = gpu_map->GetOptions().frame_shuffle_on_device;
        auto t0 = std::chrono::steady_clock::now();
        ctgn_frame_options fo;
        ctgn_frame_options_default(&fo);
        fo.frame_voxel_size = kIsAtStartup ? options.init_voxel_size : options.voxel_size;                 // :339-340
        // the keypoint voxel TryRegister will ask for first (odometry.cpp:422-423, :1038-1039): sampled in the same pass
        fo.sample_voxel_size = options.sampling == sampling::GRID
                               ? (kIsAtStartup ? options.init_sample_voxel_size : options.sample_voxel_size) : -1.0;
        fo.max_num_keypoints = -1;
        fo.override_timestamps = kIndexFrame <= 1 ? 1 : 0;                                                 // :355-359
        fo.override_timestamp = frame_info.end_timestamp;
        double pose[14], tbe[2];
        ctgn_glue::pose_to_array(tr_frame, pose, tbe);
        bool prestaged = false;
        if (shuffle_on_device) {
            fo.shuffle_seed = g() | 1u;                                                                    // one draw instead of :349's n
        } else {
            // the reference's shuffle on this thread, the upload of the scan (it travels in scan order; the order is applied on the device)
            // on another one meanwhile — when the build has OpenMP threads to spare (the reference's own loops at :469,:480 use them)
            session.order.resize(n);
            ctgn_status staged = CTGN_ERR_UNSUPPORTED;
            const ctgn_view raw_in = raw, ts_in = ts;
#pragma omp parallel num_threads(std::max(2, options.ct_icp_options.ls_num_threads))
#pragma omp sections
            {
#pragma omp section
                {
                    std::iota(session.order.begin(), session.order.end(), 0u);
                    std::shuffle(session.order.begin(), session.order.end(), g);                           // :349
                    session.ms_shuffle = ctgn_glue::ms_since(t0);
                }
#pragma omp section
                {
                    if (options.ct_icp_options.ls_num_threads > 1) staged = ctgn_frame_stage(gpu_map->handle(), raw_in, ts_in, n, &fo, pose, tbe);
                }
            }
            prestaged = staged == CTGN_OK;
        }
        if (shuffle_on_device) session.ms_shuffle = ctgn_glue::ms_since(t0);

        t0 = std::chrono::steady_clock::now();
        session.sampled.resize(n);
        ctgn_frame_outputs out{};
        out.sampled_indices = session.sampled.data();
        ctgn_view raw_begin = raw;
        if (prestaged) raw_begin.base = nullptr;                                                           // "the scan ctgn_frame_stage uploaded"
        ctgn_glue::fatal_unless_ok(ctgn_frame_begin(gpu_map->handle(), raw_begin, ts, n, shuffle_on_device ? nullptr : session.order.data(), &fo,
                                                    pose, tbe, &out), gpu_map->handle());
        const size_t n1 = (size_t) out.num_sampled;
        session.ms_begin = ctgn_glue::ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        if (!shuffle_on_device) ctgn_glue::advance_like_shuffle(session.keypoints, n1, g);                 // :361
        session.ms_shuffle += ctgn_glue::ms_since(t0);

        // the host image of the sampled frame. Its world points (:371-375, the initial estimate applied) are what TryRegister hands to
        // Register; the device derives the keypoints' from the same estimate itself, so they are fetched only if an arm downstream
        // stands down and the reference's code is about to read them (ctgn_glue::fetch_initial_world_points)
        t0 = std::chrono::steady_clock::now();
        std::vector<slam::WPoint3D> frame(n1);
        session.position.resize(n);
        for (size_t k = 0; k < n1; ++k) {
            const size_t i = session.sampled[k];
            session.position[i] = (uint32_t) k;
            auto &point = frame[k];
            point.raw_point.point = ctgn_glue::point_of(raw, i);
            point.raw_point.timestamp = kIndexFrame <= 1 ? frame_info.end_timestamp : ctgn_glue::scalar_of(ts, i);
            point.world_point = point.raw_point.point;                                                     // :345
            point.index_frame = frame_info.frame_id;                                                       // :377-379
        }
        session.ms_build_frame = ctgn_glue::ms_since(t0);
        session.world_initialised = false;
        session.registered_fid = kIndexFrame;
        session.num_points = n;
        session.num_sampled = n1;
        session.active = true;
        return frame;
    }

    // TryRegister (odometry.cpp:525-601): keypoints, startup regimen, Register, frame transform. The sampled frame is the one
    // GpuInitializeFrame left on the device; `frame` is its host image.
    inline bool GpuTryRegister(ISlamMap *map, std::vector<slam::WPoint3D> &frame, const Odometry::FrameInfo &frame_info,
                               CTICPOptions &options, Odometry::RegistrationSummary &registration_summary, double sample_voxel_size,
                               AMotionModel *motion_model, const OdometryOptions &odometry_options, std::mt19937_64 &g,
                               bool callbacks_registered) {
        auto *gpu_map = ctgn_glue::frame_pipeline_of(map, odometry_options);
        if (!gpu_map) return false;
        auto &session = gpu_map->frame_session();
        if (!session.active || session.registered_fid != frame_info.registered_fid || frame.size() != session.num_sampled) return false;
        if (callbacks_registered || options.solver == ROBUST ||
            (options.solver == CERES && (options.parametrization != CONTINUOUS_TIME || options.distance != POINT_TO_PLANE)) ||
            (motion_model && !dynamic_cast<const PreviousFrameMotionModel *>(motion_model))) {
            // not this arm's: the reference's TryRegister runs on the host image of the frame, which now needs its world points
            ctgn_glue::fetch_initial_world_points(*gpu_map, frame, registration_summary.frame);
            return false;
        }

        const auto kIndexFrame = frame_info.registered_fid;
        const bool kIsAtStartup = kIndexFrame < odometry_options.init_num_frames;
        const auto start = std::chrono::steady_clock::now();
        ctgn_frame_options fo;
        ctgn_frame_options_default(&fo);
        fo.sample_voxel_size = odometry_options.sampling == sampling::GRID ? sample_voxel_size : -1.0;     // :537-547
        fo.max_num_keypoints = !kIsAtStartup && odometry_options.max_num_keypoints > 0 ? odometry_options.max_num_keypoints : -1;   // :549
        if (kIsAtStartup) {                                                                                // :561-565
            options.threshold_voxel_occupancy = 1;
            options.num_iters_icp = std::max(options.num_iters_icp, 15);
        }

        ctgn_options co;
        ctgn_robust_options ro;
        ctgn_motion_prior prior, *prior_ptr = nullptr;
        ctgn_robust_prior rprior, *rprior_ptr = nullptr;
        const bool robust_route = options.solver == CERES;
        auto *model = dynamic_cast<const PreviousFrameMotionModel *>(motion_model);
        if (robust_route) {
            ctgn_glue::robust_options_of(options, &ro);
            if (model) { ctgn_glue::robust_prior_of(*model, &rprior); rprior_ptr = &rprior; }
        } else {
            ctgn_glue::gn_options_of(options, &co);
            if (model) { ctgn_glue::gn_prior_of(*model, &prior); prior_ptr = &prior; }
        }
        double pose[14], tbe[2];
        ctgn_glue::pose_to_array(registration_summary.frame, pose, tbe);
        session.keypoints.resize(session.num_sampled);
        session.world.resize(3 * session.num_sampled);
        ctgn_frame_outputs out{};
        out.keypoint_indices = session.keypoints.data();
        out.keypoint_world_base = session.world.data();
        out.keypoint_world_stride_bytes = 3 * sizeof(double);
        out.keypoint_world_dtype = CTGN_F64;
        ctgn_summary s;
        const ctgn_status st = ctgn_frame_try_register(gpu_map->handle(), &fo, pose, tbe, robust_route ? nullptr : &co, prior_ptr,
                                                       robust_route ? &ro : nullptr, rprior_ptr, &out, &s);
        if (st == CTGN_ERR_SOLVER) throw std::runtime_error("Error During Optimization");                  // ct_icp.cpp:628-631
        if (out.num_keypoint_candidates > out.num_keypoints && !gpu_map->GetOptions().frame_shuffle_on_device) {   // :549-552
            std::vector<uint32_t> scratch;
            ctgn_glue::advance_like_shuffle(scratch, (size_t) out.num_keypoint_candidates, g);
        }
        const size_t n2 = (size_t) out.num_keypoints;
        registration_summary.sample_size = (int) n2;                                                       // :554-555
        registration_summary.logged_values["odometry_duration_sampling"] = 0.;
        if (st == CTGN_OK) ctgn_glue::array_to_pose(pose, registration_summary.frame);
        registration_summary.icp_summary = ctgn_glue::to_summary(st, s, gpu_map->handle());                // :575-579
        registration_summary.success = registration_summary.icp_summary.success;
        registration_summary.number_of_residuals = registration_summary.icp_summary.num_residuals_used;
        registration_summary.logged_values["odometry_gpu_try_register"] =
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count();
        if (!registration_summary.success) return true;                                                    // :584-587

        // :589-595 (the frame's world points under the new poses) is GpuUndistortFrame's work: DoRegister transforms the same points
        // with the same poses again at :480-486. The keypoints with the world points the registration left them with (:597):
        registration_summary.keypoints.resize(n2);
        for (size_t k = 0; k < n2; ++k) {
            auto &point = registration_summary.keypoints[k];
            point = frame[session.position[session.keypoints[k]]];
            point.world_point = Eigen::Vector3d(session.world[3 * k], session.world[3 * k + 1], session.world[3 * k + 2]);
        }
        return true;
    }

    // The two undistortion loops of DoRegister (odometry.cpp:461-486).
    inline bool GpuUndistortFrame(ISlamMap *map, const slam::PointCloud &const_frame, const Odometry::FrameInfo &frame_info,
                                  const std::vector<slam::WPoint3D> &frame, Odometry::RegistrationSummary &summary, int num_threads) {
        auto *gpu_map = dynamic_cast<GpuVoxelMap *>(map);
        if (!gpu_map) return false;
        auto &session = gpu_map->frame_session();
        if (!session.active || session.registered_fid != frame_info.registered_fid || frame.size() != session.num_sampled ||
            const_frame.size() != session.num_points)
            return false;
        const size_t n = session.num_points;
        auto t0 = std::chrono::steady_clock::now();
        ctgn_view raw, ts;
        if (!ctgn_glue::view_of(const_frame.XYZConst<double>(), &raw) || !ctgn_glue::view_of(const_frame.TimestampsProxy<double>(), &ts)) return false;
        if (session.rows_capacity < n) {                   // page-locked rows for the read-back (kept from frame to frame)
            if (session.rows) ctgn_host_free(gpu_map->handle(), session.rows);
            session.rows = nullptr;
            session.rows_capacity = 0;
            void *p = nullptr;
            if (ctgn_host_alloc(gpu_map->handle(), (n + n / 4) * 3 * sizeof(double), &p) == CTGN_OK) {
                session.rows = static_cast<double *>(p);
                session.rows_capacity = n + n / 4;
            } else {
                session.world.resize(3 * n);               // plain memory does too (the library copies instead of the DMA engine)
            }
        }
        double *rows = session.rows ? session.rows : session.world.data();
        summary.corrected_points = frame;                                                                  // :462
        const auto &begin_pose = summary.frame.begin_pose;
        const auto &end_pose = summary.frame.end_pose;
        // the first two frames were staged with every timestamp at the end of the sweep (:355-359), but all_corrected_points carries the
        // points' own timestamps (:473): those two frames take the reference's loop for it
        const bool all_on_device = session.registered_fid > 1;
        double pose[14], tbe[2];
        ctgn_glue::pose_to_array(summary.frame, pose, tbe);
        ctgn_frame_outputs out{};
        if (all_on_device) {
            out.all_world_base = rows;
            out.all_world_stride_bytes = 3 * sizeof(double);
            out.all_world_dtype = CTGN_F64;
        }
        if (!summary.corrected_points.empty()) {
            out.sampled_world_base = summary.corrected_points[0].world_point.data();                       // :480-486
            out.sampled_world_stride_bytes = sizeof(slam::WPoint3D);
            out.sampled_world_dtype = CTGN_F64;
        }
        ctgn_status st = CTGN_OK;
        double ms_device = 0.;
        // one thread drives the device (every scan point's world point arrives as rows), another sizes all_corrected_points meanwhile; then
        // the team fills the records: raw point, timestamp, frame id (:472-474) and the world point from the rows
#pragma omp parallel num_threads(num_threads)
        {
#pragma omp single nowait
            {
                const auto t1 = std::chrono::steady_clock::now();
                st = ctgn_frame_undistort(gpu_map->handle(), pose, tbe, &out);
                ms_device = ctgn_glue::ms_since(t1);
            }
#pragma omp single nowait
            summary.all_corrected_points.resize(n);                                                        // :463
#pragma omp barrier
#pragma omp for schedule(static)
            for (auto i = 0; i < n; ++i) {                                                                 // :470-478
                auto &point = summary.all_corrected_points[i];
                point.RawPoint() = ctgn_glue::point_of(raw, i);
                point.Timestamp() = ctgn_glue::scalar_of(ts, i);
                point.index_frame = frame_info.frame_id;
                if (all_on_device) point.WorldPoint() = Eigen::Vector3d(rows[3 * i], rows[3 * i + 1], rows[3 * i + 2]);
                else point.WorldPoint() = begin_pose.ContinuousTransform(point.RawPoint(), end_pose, point.Timestamp());
            }
        }
        ctgn_glue::fatal_unless_ok(st, gpu_map->handle());
        session.undistorted = true;
        session.ms_undistort = ms_device;
        session.ms_fill = ctgn_glue::ms_since(t0);
        summary.logged_values["odometry_gpu_shuffle"] = session.ms_shuffle;
        summary.logged_values["odometry_gpu_frame_begin"] = session.ms_begin;
        summary.logged_values["odometry_gpu_build_frame"] = session.ms_build_frame;
        summary.logged_values["odometry_gpu_undistort_call"] = session.ms_undistort;
        summary.logged_values["odometry_gpu_undistort_arm"] = session.ms_fill;
        return true;
    }

    // The map half of UpdateMap (odometry.cpp:936-952): far-voxel eviction round the new location, then the corrected sampled frame —
    // taken from where GpuUndistortFrame left it on the device.
    inline bool GpuUpdateMap(ISlamMap *map, const Eigen::Vector3d &location, double max_distance, bool add_points) {
        auto *gpu_map = dynamic_cast<GpuVoxelMap *>(map);
        if (!gpu_map) return false;
        auto &session = gpu_map->frame_session();
        if (!session.active || !session.undistorted) return false;
        const double loc[3] = {location[0], location[1], location[2]};
        ctgn_glue::fatal_unless_ok(ctgn_frame_update_map(gpu_map->handle(), loc, max_distance, add_points ? 1 : 0, nullptr), gpu_map->handle());
        session.active = false;
        return true;
    }

} // namespace ct_icp

#endif //CT_ICP_ODOMETRY_GPU_ARM_H
