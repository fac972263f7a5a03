// adapter_check — compiles the C++ adapter against libctgn.so and runs one registration through it.
// Reference-style caller code (cf. Odometry::TryRegister, reference src/ct_icp/odometry.cpp:573-579):
//     auto summary = registration.Register(*map, keypoints, frame, &motion_model);
// Output (one line): "adapter ok n_used=<int> iters=<int> tr=<x y z> ..." or "adapter no-device" on a box without a GPU.
#include <cmath>
#include <cstdio>
#include <random>

#include "ct_icp_gpu.hpp"

using namespace ct_icp_gpu;

int main() {
    GpuVoxelMap::Options mo;
    mo.resolutions = {{0.5, 0.05, 20}};
    mo.default_radius = 0.8;
    try {
        GpuVoxelMap map(mo);
        // a closed 8 m box sampled on its six faces
        std::mt19937_64 g(5489);
        std::uniform_real_distribution<double> u(-4.0, 4.0);
        std::vector<double> pts;
        for (int i = 0; i < 60000; ++i) {
            double p[3] = {u(g), u(g), u(g)};
            int ax = i % 3;
            p[ax] = (i / 3) % 2 ? 4.0 : -4.0;
            pts.insert(pts.end(), p, p + 3);
        }
        map.InsertPoints(pts.data(), pts.size() / 3);
        // keypoints: points of the same faces seen from a sensor displaced by a small rigid motion
        const double shift[3] = {0.03, -0.02, 0.01};
        std::vector<WPoint3D> kps;
        for (int i = 0; i < 3000; ++i) {
            WPoint3D w{};
            double p[3] = {u(g) * 0.8, u(g) * 0.8, u(g) * 0.8};
            int ax = i % 3;
            p[ax] = (i / 3) % 2 ? 4.0 : -4.0;
            for (int c = 0; c < 3; ++c) { w.raw_point[c] = p[c] - shift[c]; w.world_point[c] = w.raw_point[c]; }
            w.timestamp = (double) i / 3000.0;
            kps.push_back(w);
        }
        TrajectoryFrame frame;
        frame.begin_pose.dest_timestamp = 0.0;
        frame.end_pose.dest_timestamp = 1.0;
        CT_ICP_Registration reg;
        reg.Options().solver = GN;
        reg.Options().num_iters_icp = 10;
        reg.Options().debug_print = false;
        ICPSummary s = reg.Register(map, kps, frame, nullptr);
        double err = 0;
        for (int c = 0; c < 3; ++c) err = std::fmax(err, std::fabs(frame.end_pose.pose.tr[c] - shift[c]));
        std::printf("adapter %s n_used=%d iters=%d tr=%.6f %.6f %.6f err=%.2e map_points=%zu\n",
                    (s.success && err < 1e-6) ? "ok" : "FAIL", s.num_residuals_used, s.num_iters, frame.end_pose.pose.tr[0],
                    frame.end_pose.pose.tr[1], frame.end_pose.pose.tr[2], err, map.NumPoints());
        // raw-points insertion with poses + the in-place query spellings
        {
            std::vector<WPoint3D> extra(kps.begin(), kps.begin() + 500);
            std::vector<size_t> kept;
            map.InsertPointCloud(extra, frame.begin_pose, frame.end_pose, kept);
            Neighborhood nb;
            map.ComputeNeighborhoodInPlace(extra[0].world_point, 5, nb);
            if (nb.size() == 0) { std::printf("adapter FAIL: empty neighbourhood\n"); return 1; }
        }
        // the steps either side of the path: both samplers and the undistortion loop
        bool side_ok = true;
        {
            std::vector<WPoint3D> sampled;
            grid_sampling(map, kps, sampled, 1.0);
            std::vector<WPoint3D> copy = kps;
            sub_sample_frame(map, copy, 1.0);
            std::vector<size_t> adaptive = AdaptiveSamplePointsInGrid(map, kps, AdaptiveGridSamplingOptions());
            side_ok = !sampled.empty() && sampled.size() < kps.size() && copy.size() == sampled.size() && !adaptive.empty() &&
                      adaptive.size() < kps.size();
            TransformFrame(map, copy, frame);
            double dmax = 0;
            for (const WPoint3D &p : copy)
                for (int c = 0; c < 3; ++c) dmax = std::fmax(dmax, std::fabs(p.world_point[c] - p.raw_point[c] - (1.0 - p.timestamp) * frame.begin_pose.pose.tr[c] -
                                                     p.timestamp * frame.end_pose.pose.tr[c]));
            side_ok = side_ok && dmax < 1e-5;                     // the optimised rotations are the identity to ~1e-7
            std::printf("adapter-sampling %s grid=%zu adaptive=%zu undistort-err=%.1e\n", side_ok ? "ok" : "FAIL", sampled.size(),
                        adaptive.size(), dmax);
        }
        // one whole frame with the scan resident on the device, on a device-maintained map built from the same points: it must agree
        // with the stage-by-stage spelling above (same kernels, same inputs)
        {
            GpuVoxelMap dmap(mo);
            if (ctgn_map_set_update_mode(dmap.handle(), 1) != CTGN_OK) { std::printf("adapter FAIL: update mode\n"); return 1; }
            dmap.InsertPoints(pts.data(), pts.size() / 3);
            TrajectoryFrame f3;
            f3.begin_pose.dest_timestamp = 0.0;
            f3.end_pose.dest_timestamp = 1.0;
            FrameOptions fopt;
            fopt.voxel_size = 0.0;                                // every point of `kps` stays ...
            fopt.sample_voxel_size = 0.0;                         // ... and is a keypoint: the same registration as `s`
            std::vector<WPoint3D> all, corrected;
            std::vector<uint32_t> kp_idx;
            CTICPOptions go = reg.Options();
            ICPSummary s3 = RegisterFrame(dmap, go, fopt, kps, f3, nullptr, &all, &corrected, &kp_idx);
            bool same = s3.success && s3.num_iters == s.num_iters && s3.num_residuals_used == s.num_residuals_used;
            for (int c = 0; c < 3; ++c) same = same && f3.end_pose.pose.tr[c] == frame.end_pose.pose.tr[c];
            const size_t before = dmap.NumPoints();
            UpdateMapFromFrame(dmap, f3.end_pose.pose.tr, 100.0, true);
            std::printf("adapter-frame %s sampled=%zu keypoints=%zu map_points %zu -> %zu\n", same ? "ok" : "FAIL", corrected.size(),
                        kp_idx.size(), before, dmap.NumPoints());
            side_ok = side_ok && same && all.size() == kps.size() && corrected.size() == kps.size() && dmap.NumPoints() >= before;
        }
        // the `case CERES:` arm on the same map and keypoints
        TrajectoryFrame frame2;
        frame2.begin_pose.dest_timestamp = 0.0;
        frame2.end_pose.dest_timestamp = 1.0;
        reg.Options().solver = CERES;
        reg.Options().ls_max_num_iters = 5;
        reg.Options().threshold_orientation_norm = 1e-6;
        reg.Options().threshold_translation_norm = 1e-8;
        ICPSummary s2 = reg.Register(map, kps, frame2, nullptr);
        double err2 = 0;
        for (int c = 0; c < 3; ++c) err2 = std::fmax(err2, std::fabs(frame2.end_pose.pose.tr[c] - shift[c]));
        std::printf("adapter-ceres %s n_res=%d iters=%d err=%.2e\n", (s2.success && err2 < 1e-5) ? "ok" : "FAIL",
                    s2.num_residuals_used, s2.num_iters, err2);
        return (s.success && err < 1e-6 && s2.success && err2 < 1e-5 && side_ok) ? 0 : 1;
    } catch (const std::exception &e) {
        std::printf("adapter no-device (%s)\n", e.what());
        return 0;
    }
}
