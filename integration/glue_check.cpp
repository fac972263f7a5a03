// integration/glue_check.cpp — the drop-in, end to end, inside the REFERENCE'S OWN CODE: this program is linked with the reference's
// sources (oracle/Makefile: ct_icp.cpp with the two one-line insertions of gn_gpu_arm.h, map.cpp, motion_model.cpp, SlamCore ...)
// and with libctgn.so. It calls ct_icp::CT_ICP_Registration::Register (ct_icp.cpp:1026-1038) — the entry point Odometry::TryRegister
// uses — once on the reference's MultipleResolutionVoxelMap (its CPU loop runs) and once on GpuVoxelMap (the inserted arm hands the
// call to the GPU), for `solver: GN` and `solver: CERES`, on vector<WPoint3D> keypoints (FLOAT64 views, world point at offset 32 of a
// 64-byte item) and on a slam::PointCloud with FLOAT32 fields, and compares.
// Output: one line per case, "glue <case> ok|FAIL ..."; "glue no-device" where ctgn_create finds no gfx950 (CPU container).
// Test infrastructure (tests/test_integration_glue.py); third-party arithmetic underneath the reference is oracle/shims/.
#include <cmath>
#include <cstdio>
#include <random>

#include <ct_icp/ct_icp.h>
#include <ct_icp/gpu_map.h>
#include <ct_icp/motion_model.h>

namespace {
    double pose_diff(const ct_icp::TrajectoryFrame &a, const ct_icp::TrajectoryFrame &b) {
        double d = 0;
        for (int i = 0; i < 7; ++i) {
            d = std::fmax(d, std::fabs(a.begin_pose.pose[i] - b.begin_pose.pose[i]));
            d = std::fmax(d, std::fabs(a.end_pose.pose[i] - b.end_pose.pose[i]));
        }
        return d;
    }

    slam::PointCloudPtr cloud_of(const std::vector<Eigen::Vector3d> &pts) {
        auto pc = slam::PointCloud::DefaultXYZPtr<double>();
        pc->resize(pts.size());
        pc->SetWorldPointsField(slam::PointCloud::Field{pc->GetXYZField()});
        auto xyz = pc->XYZ<double>();
        for (size_t i = 0; i < pts.size(); ++i) xyz[i] = pts[i];
        return pc;
    }
}

int main() {
    using namespace ct_icp;
    // a closed 8 m box sampled on its six faces (the scene of the reference's integration test, test/integration/testint_utils.h:39-96)
    std::mt19937_64 g(5489);
    std::uniform_real_distribution<double> u(-4.0, 4.0);
    std::vector<Eigen::Vector3d> pts;
    for (int i = 0; i < 60000; ++i) {
        Eigen::Vector3d p(u(g), u(g), u(g));
        p[i % 3] = (i / 3) % 2 ? 4.0 : -4.0;
        pts.push_back(p);
    }
    MultipleResolutionVoxelMap::Options cpu_options;
    cpu_options.resolutions = {MultipleResolutionVoxelMap::ResolutionParam{0.5, 0.05, 20}};
    cpu_options.default_radius = 0.8;
    GpuVoxelMap::Options gpu_options;
    gpu_options.resolutions = cpu_options.resolutions;
    gpu_options.default_radius = cpu_options.default_radius;

    std::unique_ptr<GpuVoxelMap> gpu_map;
    try {
        gpu_map = std::make_unique<GpuVoxelMap>(gpu_options);
    } catch (const std::exception &e) {
        std::printf("glue no-device (%s)\n", e.what());
        return 0;
    }
    MultipleResolutionVoxelMap cpu_map(cpu_options);
    auto cloud = cloud_of(pts);
    std::vector<size_t> kept_cpu, kept_gpu;
    cpu_map.InsertPointCloud(*cloud, kept_cpu);
    gpu_map->InsertPointCloud(*cloud, kept_gpu);
    bool all_ok = cpu_map.NumPoints() == gpu_map->NumPoints();
    std::printf("glue insert %s cpu=%zu gpu=%zu points\n", all_ok ? "ok" : "FAIL", cpu_map.NumPoints(), gpu_map->NumPoints());
    {   // ISlamMap queries through the interface
        const ISlamMap &a = cpu_map, &b = *gpu_map;
        size_t bad = 0;
        for (int i = 0; i < 200; ++i) {
            auto na = a.ComputeNeighborhood(pts[(size_t) i * 7], 20), nb = b.ComputeNeighborhood(pts[(size_t) i * 7], 20);
            bad += na.points.size() != nb.points.size();
            for (size_t j = 0; j < na.points.size() && j < nb.points.size(); ++j) bad += !(na.points[j] == nb.points[j]);
        }
        std::printf("glue neighbourhoods %s (%zu mismatches)\n", bad == 0 ? "ok" : "FAIL", bad);
        all_ok = all_ok && bad == 0;
    }

    // keypoints: points of the same faces seen from a sensor displaced by a small rigid motion
    const Eigen::Vector3d shift(0.03, -0.02, 0.01);
    std::vector<slam::WPoint3D> keypoints;
    for (int i = 0; i < 3000; ++i) {
        slam::WPoint3D w;
        Eigen::Vector3d p(u(g) * 0.8, u(g) * 0.8, u(g) * 0.8);
        p[i % 3] = (i / 3) % 2 ? 4.0 : -4.0;
        w.RawPoint() = p - shift;
        w.WorldPoint() = w.RawPoint();
        w.Timestamp() = (double) i / 3000.0;
        keypoints.push_back(w);
    }
    PreviousFrameMotionModel model;
    TrajectoryFrame previous;
    previous.begin_pose.dest_timestamp = -1.0; previous.end_pose.dest_timestamp = 0.0;
    model.UpdateState(previous, 0);

    for (int route = 0; route < 2; ++route) {
        CT_ICP_Registration registration;
        registration.Options().solver = route == 0 ? GN : CERES;
        registration.Options().num_iters_icp = route == 0 ? 10 : 4;
        registration.Options().ls_max_num_iters = 5;
        registration.Options().ls_num_threads = 1;
        registration.Options().debug_print = false;
        TrajectoryFrame frames[2];
        ICPSummary summaries[2];
        std::vector<slam::WPoint3D> kps[2] = {keypoints, keypoints};
        for (int m = 0; m < 2; ++m) {
            frames[m].begin_pose.dest_timestamp = 0.0;
            frames[m].end_pose.dest_timestamp = 1.0;
            const ISlamMap &map = m == 0 ? static_cast<const ISlamMap &>(cpu_map) : static_cast<const ISlamMap &>(*gpu_map);
            summaries[m] = registration.Register(map, kps[m], frames[m], &model);
        }
        double wdiff = 0;
        for (size_t i = 0; i < keypoints.size(); ++i) wdiff = std::fmax(wdiff, (kps[0][i].WorldPoint() - kps[1][i].WorldPoint()).norm());
        const double pd = pose_diff(frames[0], frames[1]);
        const bool ok = summaries[0].success && summaries[1].success && summaries[0].num_residuals_used == summaries[1].num_residuals_used &&
                        pd < 1e-7 && wdiff < 1e-7 && (frames[1].end_pose.TrRef() - shift).norm() < 1e-3;
        std::printf("glue %s %s n_used cpu=%d gpu=%d pose_diff=%.2e world_diff=%.2e gpu avg ms: iter %.4f neighborhood %.4f solve %.4f\n",
                    route == 0 ? "GN" : "CERES", ok ? "ok" : "FAIL", summaries[0].num_residuals_used, summaries[1].num_residuals_used, pd, wdiff,
                    summaries[1].avg_duration_iter, summaries[1].avg_duration_neighborhood, summaries[1].avg_duration_solve);
        all_ok = all_ok && ok;
    }

    {   // the slam::PointCloud overload (ct_icp.cpp:1040-1053) on FLOAT32 raw / world points: the views arrive with
        // src_property_type FLOAT32 and the GPU arm reads / writes them as such
        auto pc = slam::PointCloud::DefaultXYZPtr<float>();
        pc->resize(keypoints.size());
        pc->SetRawPointsField(slam::PointCloud::Field{pc->GetXYZField()});
        pc->AddDefaultWorldPointsField();
        pc->AddDefaultTimestampsField();
        auto raw = pc->RawPointsProxy<Eigen::Vector3d>();
        auto world = pc->WorldPointsProxy<Eigen::Vector3d>();
        auto ts = pc->TimestampsProxy<double>();
        for (size_t i = 0; i < keypoints.size(); ++i) {
            raw[i] = keypoints[i].RawPoint();
            world[i] = keypoints[i].RawPoint();
            ts[i] = keypoints[i].Timestamp();
        }
        CT_ICP_Registration registration;
        registration.Options().solver = GN;
        registration.Options().num_iters_icp = 10;
        registration.Options().debug_print = false;
        TrajectoryFrame frame;
        frame.begin_pose.dest_timestamp = 0.0;
        frame.end_pose.dest_timestamp = 1.0;
        ICPSummary s = registration.Register(*gpu_map, *pc, frame, nullptr);
        const double err = (frame.end_pose.TrRef() - shift).norm();
        const bool ok = s.success && err < 1e-3;
        std::printf("glue PointCloud(float32 views) %s n_used=%d err=%.2e\n", ok ? "ok" : "FAIL", s.num_residuals_used, err);
        all_ok = all_ok && ok;
    }
    std::printf("glue %s\n", all_ok ? "ALL OK" : "FAILED");
    return all_ok ? 0 : 1;
}
