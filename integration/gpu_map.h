// integration/gpu_map.h — the file a maintainer drops into the reference tree as include/ct_icp/gpu_map.h:
// `GpuVoxelMap : ct_icp::ISlamMap`, the reference's map interface (include/ct_icp/map.h:14-83, include/SlamCore/experimental/map.h:25-45)
// served by libctgn.so (include/ctgn.h). Registered under `map_type: GPU_VOXEL_HASHMAP` (src/ct_icp/map.cpp:68-77).
//
// Compiled verbatim against the reference's own headers by tests/test_integration_glue.py (with the third-party shims of
// oracle/shims/, since this image has no Eigen / glog / ...), and linked with the reference's own ct_icp.cpp (patched by one line per
// solver arm, see gn_gpu_arm.h) into oracle/_ref/glue_check, which registers a scan on the GPU through the reference's
// CT_ICP_Registration::Register and compares with the reference's CPU map.
#ifndef CT_ICP_GPU_MAP_H
#define CT_ICP_GPU_MAP_H

#include <atomic>
#include <type_traits>
#include <cstdint>
#include <vector>

#include <yaml-cpp/yaml.h>

#include <ct_icp/map.h>
#include <SlamCore/config_utils.h>
#include <ctgn.h>

namespace ct_icp {

    namespace ctgn_glue {
        // A slam::ProxyView is {item_buffer.view_data_ptr, offset_in_item, item_size, src_property_type} (SlamCore/data/view.h:98-116,
        // 186-189): element i lives at view_data_ptr + i * item_size + offset_in_item — for vector<WPoint3D> the world point sits at
        // offset 32 of a 64-byte item. The C ABI takes exactly that as a strided view. Only FLOAT32 / FLOAT64 sources have a device
        // path; the other property types (proxy_ref.h:84-126 casts them on access) make the callers keep the reference's CPU loop.
        template<typename T>
        inline bool view_of(const slam::ProxyView<T> &v, ctgn_view *out) {
            if (v.src_property_type != slam::FLOAT32 && v.src_property_type != slam::FLOAT64) return false;
            out->base = v.item_buffer.view_data_ptr + v.offset_in_item;
            out->stride_bytes = (size_t) v.item_size;
            out->dtype = v.src_property_type == slam::FLOAT64 ? CTGN_F64 : CTGN_F32;
            out->_pad = 0;
            return true;
        }

        // ... and a typed slam::View<float / double> (view.h:20-62) is the same triple with the type fixed
        template<typename T>
        inline bool view_of(const slam::View<T> &v, ctgn_view *out) {
            if (!std::is_same<T, double>::value && !std::is_same<T, float>::value) return false;
            out->base = v.item_buffer.view_data_ptr + v.offset_in_item;
            out->stride_bytes = (size_t) v.item_size;
            out->dtype = std::is_same<T, double>::value ? CTGN_F64 : CTGN_F32;
            out->_pad = 0;
            return true;
        }

        // how many GpuVoxelMaps with `frame_pipeline` are alive in the process: the one arm that has no Odometry to ask
        // (GpuFrameTimeRange, odometry_gpu_arm.h — called from a file-scope lambda of odometry.cpp) stands down at zero
        inline std::atomic<int> &frame_pipeline_maps() {
            static std::atomic<int> count{0};
            return count;
        }
    }

    class GpuVoxelMap : public ISlamMap {
    public:
        // Same fields and YAML keys as MultipleResolutionVoxelMap::Options (map.h:109-134, src/ct_icp/map.cpp:32-65) + `device`. Derived from
        // IMapOptions itself: MultipleResolutionVoxelMap::Options::MakeMapFromOptions is `final` (map.h:130), and the factory call is how
        // Odometry::Odometry / Odometry::Reset come by their map (src/ct_icp/odometry.cpp:700,973).
        struct Options : public IMapOptions {
            std::vector<MultipleResolutionVoxelMap::ResolutionParam> resolutions = MultipleResolutionVoxelMap::Options().resolutions;
            size_t max_frames_to_keep = 100;                                // kept for the YAML loader; the device map keeps no frame ids
            double default_radius = 0.8;
            int device = 0;
            bool device_updates = true;                                     // insert / evict rules run on the GPU (ctgn_map_set_update_mode)
            bool frame_pipeline = true;                                     // Odometry::DoRegister's scan-sized loops run on the GPU too, where
                                                                            // src/ct_icp/odometry.cpp carries the arms of odometry_gpu_arm.h
            bool frame_shuffle_on_device = false;                           // ... and so does the shuffle in front of sub_sample_frame: a keyed
                                                                            // permutation made on the GPU (seeded from Odometry's engine) instead
                                                                            // of std::shuffle on the host (~1 ms for a 132 k-point scan). Off: the
                                                                            // sampled frame is point for point the one an un-armed run keeps

            static std::string Type() { return "GPU_VOXEL_HASHMAP"; }

            std::string GetType() const override { return Type(); }

            inline std::shared_ptr<ISlamMap> MakeMapFromOptions() const final { return std::make_shared<GpuVoxelMap>(*this); }
        };

        explicit GpuVoxelMap(const Options &options) : options_(options) {
            ctgn_map_options mo;
            ctgn_map_options_default(&mo);
            SLAM_CHECK_STREAM(options.resolutions.size() <= CTGN_MAX_RESOLUTIONS, "too many resolutions for libctgn");
            mo.num_resolutions = (int32_t) options.resolutions.size();
            mo.device = options.device;
            mo.default_radius = options.default_radius;
            for (size_t i = 0; i < options.resolutions.size(); ++i) {
                mo.resolutions[i].resolution = options.resolutions[i].resolution;
                mo.resolutions[i].min_distance_between_points = options.resolutions[i].min_distance_between_points;
                mo.resolutions[i].max_num_points = options.resolutions[i].max_num_points;
            }
            const ctgn_status st = ctgn_create(&mo, &handle_);
            SLAM_CHECK_STREAM(st == CTGN_OK, "libctgn: " << ctgn_status_string(st));
            if (options.device_updates) ctgn_map_set_update_mode(handle_, 1);
            if (options.frame_pipeline && options.device_updates) ctgn_glue::frame_pipeline_maps()++;
        }

        ~GpuVoxelMap() override {
            if (options_.frame_pipeline && options_.device_updates) ctgn_glue::frame_pipeline_maps()--;
            if (session_.rows) ctgn_host_free(handle_, session_.rows);
            ctgn_destroy(handle_);
        }

        GpuVoxelMap(const GpuVoxelMap &) = delete;

        ctgn_handle handle() const { return handle_; }

        const Options &GetOptions() const { return options_; }

        // What the arms of odometry_gpu_arm.h hand from one step of Odometry::DoRegister to the next: the scan InitializeFrame's arm left
        // on the device (ctgn_frame_begin) and the index lists that tie the host's vectors to it. Scratch vectors keep their capacity from
        // frame to frame (a 132 k-point scan would otherwise cost a megabyte of page faults per frame).
        struct FrameSession {
            bool active = false;               // a sampled scan of THIS frame is resident
            bool undistorted = false;          // ... and ctgn_frame_undistort has run on it (the map update may take it)
            int registered_fid = -1;
            size_t num_points = 0, num_sampled = 0;
            std::vector<uint32_t> order;       // the first shuffle (odometry.cpp:349), as indices
            std::vector<uint32_t> sampled;     // scan index of sampled point k
            std::vector<uint32_t> position;    // position[sampled[k]] = k (only those entries are meaningful)
            std::vector<uint32_t> keypoints;   // scan index of keypoint k
            std::vector<double> world;         // world points of a read-back, x y z rows
            double *rows = nullptr;            // page-locked x y z rows for every scan point's world point (ctgn_host_alloc): the undistortion
            size_t rows_capacity = 0;          // lands there by DMA, the fill loop of all_corrected_points reads it
            bool world_initialised = false;    // the host image of the sampled frame carries its world points under the initial estimate
            // host milliseconds of the arms' steps, for RegistrationSummary::logged_values (odometry_gpu_*)
            double ms_shuffle = 0, ms_begin = 0, ms_build_frame = 0, ms_fill = 0, ms_undistort = 0;
        };

        FrameSession &frame_session() { return session_; }

        ////////////////////////////////////////////////////////////////////////////////////////////////////////////////
        /// UPDATE API (map.h:25-29, SlamCore/experimental/map.h:32-39)
        ////////////////////////////////////////////////////////////////////////////////////////////////////////////////

        void InsertPointCloud(const slam::PointCloud &pointcloud, const std::vector<slam::Pose> &frame_poses,
                              std::vector<size_t> &out_indices) override {
            SLAM_CHECK_STREAM(!frame_poses.empty(), "the poses are empty");
            // world points exactly as MultipleResolutionVoxelMap::InsertPointCloud derives them (map.h:156-184)
            auto pc = pointcloud.DeepCopyPtr();
            pc->RegisterFieldsFromSchema();
            if (!pc->HasWorldPoints()) {
                SLAM_CHECK_STREAM(pc->HasRawPoints(), "The input point cloud does not have raw points defined");
                pc->AddDefaultWorldPointsField();
                auto trajectory = slam::LinearContinuousTrajectory::Create(std::vector<slam::Pose>(frame_poses));
                if (pc->HasTimestamps() && trajectory.Poses().size() >= 2)
                    pc->RawPointsToWorldPoints(trajectory);
                else
                    pc->RawPointsToWorldPoints(trajectory.Poses().front().pose);
            }
            auto xyz = pc->WorldPointsProxy<Eigen::Vector3d>();
            const size_t n = xyz.size();
            std::vector<uint8_t> inserted(n, 0);
            ctgn_view view;
            if (ctgn_glue::view_of(xyz, &view)) {
                Check(ctgn_map_insert(handle_, view.base, view.stride_bytes, view.dtype, n, inserted.data()));
            } else {                                                        // exotic source type: cast on the host, as proxy_ref.h does
                std::vector<double> buf(3 * n);
                for (size_t i = 0; i < n; ++i) {
                    Eigen::Vector3d p = xyz[i];
                    buf[3 * i] = p[0]; buf[3 * i + 1] = p[1]; buf[3 * i + 2] = p[2];
                }
                Check(ctgn_map_insert(handle_, buf.data(), 24, CTGN_F64, n, inserted.data()));
            }
            for (size_t i = 0; i < n; ++i) if (inserted[i]) out_indices.push_back(i);
        }

        void InsertPointCloud(const slam::PointCloud &cloud, std::vector<size_t> &out_selected_points) override {
            InsertPointCloud(cloud, {slam::Pose()}, out_selected_points);
        }

        void ClearMap() override { Check(ctgn_map_clear(handle_)); }

        void RemoveElementsFarFromLocation(const Eigen::Vector3d &location, double distance) override {
            const double loc[3] = {location[0], location[1], location[2]};
            Check(ctgn_map_remove_far(handle_, loc, distance));
        }

        ////////////////////////////////////////////////////////////////////////////////////////////////////////////////
        /// EXPORT API (map.h:31-43)
        ////////////////////////////////////////////////////////////////////////////////////////////////////////////////

        size_t NumPoints() const override {
            uint64_t n = 0;
            Check(ctgn_map_num_points(handle_, &n));
            return (size_t) n;
        }

        slam::PointCloudPtr MapAsPointCloud() const override {
            uint64_t n = 0;
            Check(ctgn_map_export(handle_, 0, nullptr, 0, &n));
            std::vector<double> buf(3 * (size_t) n);
            Check(ctgn_map_export(handle_, 0, buf.data(), n, &n));
            auto pc = slam::PointCloud::DefaultXYZPtr<double>();
            pc->resize((size_t) n);
            pc->SetWorldPointsField(slam::PointCloud::Field{pc->GetXYZField()});
            auto xyz = pc->XYZ<double>();
            for (size_t i = 0; i < (size_t) n; ++i) xyz[i] = Eigen::Vector3d(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]);
            return pc;
        }

        ////////////////////////////////////////////////////////////////////////////////////////////////////////////////
        /// QUERY API (map.h:55-82, SlamCore/experimental/map.h:40-45): batched on the GPU, farthest neighbour first (map.h:508-513)
        ////////////////////////////////////////////////////////////////////////////////////////////////////////////////

        void RadiusSearchInPlace(const Eigen::Vector3d &query, slam::Neighborhood &neighborhood, double radius,
                                 int max_num_neighbors, bool nearest_neighbors, Eigen::Vector3d *sensor_location) const override {
            std::vector<slam::Neighborhood> out = Search({query}, {radius}, max_num_neighbors);
            neighborhood.points = std::move(out.front().points);
        }

        slam::Neighborhood RadiusSearch(const Eigen::Vector3d &query, double radius, int max_num_neighbors,
                                        bool nearest_neighbors, Eigen::Vector3d *sensor_location) const override {
            slam::Neighborhood neighborhood;
            RadiusSearchInPlace(query, neighborhood, radius, max_num_neighbors, nearest_neighbors, sensor_location);
            return neighborhood;
        }

        std::vector<slam::Neighborhood> ComputeNeighborhoods(const std::vector<Eigen::Vector3d> &queries,
                                                             const std::vector<double> radiuses, int max_num_neighbors,
                                                             bool nearest_neighbors, Eigen::Vector3d *sensor_location) const override {
            SLAM_CHECK_STREAM(radiuses.size() == queries.size(), "Invalid Parameters, size of queries and radiuses do not match");
            return Search(queries, radiuses, max_num_neighbors);
        }

        void ComputeNeighborhoodInPlace(const Eigen::Vector3d &query, int max_num_neighbors,
                                        slam::Neighborhood &neighborhood) const override {
            RadiusSearchInPlace(query, neighborhood, options_.default_radius, max_num_neighbors, true, nullptr);
        }

        std::vector<slam::Neighborhood> ComputeNeighborhoods(const std::vector<Eigen::Vector3d> &queries,
                                                             int max_num_neighbors) const override {
            return Search(queries, std::vector<double>(queries.size(), options_.default_radius), max_num_neighbors);
        }

    private:
        // one ctgn_map_radius_search launch per distinct radius (the usual case: a single radius for the whole batch)
        std::vector<slam::Neighborhood> Search(const std::vector<Eigen::Vector3d> &queries, const std::vector<double> &radiuses,
                                               int max_num_neighbors) const {
            const int k = max_num_neighbors < 0 ? CTGN_MAX_NEIGHBORS : max_num_neighbors;
            SLAM_CHECK_STREAM(k >= 1 && k <= CTGN_MAX_NEIGHBORS, "libctgn searches at most " << CTGN_MAX_NEIGHBORS << " neighbours");
            std::vector<slam::Neighborhood> result(queries.size());
            std::vector<bool> done(queries.size(), false);
            for (size_t first = 0; first < queries.size(); ++first) {
                if (done[first]) continue;
                std::vector<size_t> batch;
                for (size_t i = first; i < queries.size(); ++i)
                    if (!done[i] && radiuses[i] == radiuses[first]) { batch.push_back(i); done[i] = true; }
                std::vector<double> q(3 * batch.size()), out(3 * (size_t) k * batch.size());
                std::vector<int32_t> count(batch.size());
                for (size_t j = 0; j < batch.size(); ++j)
                    for (int c = 0; c < 3; ++c) q[3 * j + c] = queries[batch[j]][c];
                Check(ctgn_map_radius_search(handle_, q.data(), batch.size(), radiuses[first], k, out.data(), count.data()));
                for (size_t j = 0; j < batch.size(); ++j) {
                    auto &points = result[batch[j]].points;
                    points.resize((size_t) count[j]);
                    for (int32_t m = 0; m < count[j]; ++m) {
                        const double *p = &out[((size_t) j * k + (size_t) m) * 3];
                        points[(size_t) m] = Eigen::Vector3d(p[0], p[1], p[2]);
                    }
                }
            }
            return result;
        }

        void Check(ctgn_status st) const {
            SLAM_CHECK_STREAM(st == CTGN_OK, "libctgn: " << ctgn_last_error(handle_));
        }

        Options options_;
        ctgn_handle handle_ = nullptr;
        FrameSession session_;
    };

    // The YAML loader of `map_type: GPU_VOXEL_HASHMAP`: the keys of MULTI_RESOLUTION_VOXEL_HASHMAP (multi_resolution_map_options_from_yaml,
    // src/ct_icp/map.cpp:32-65) + `device`, `device_updates`, `frame_pipeline`. The selector line in yaml_to_map_options (src/ct_icp/map.cpp:68-77):
    //     if (map_type == GpuVoxelMap::Options::Type()) return gpu_map_options_from_yaml(node);
    inline std::shared_ptr<ct_icp::IMapOptions> gpu_map_options_from_yaml(const YAML::Node &node) {
        auto map_options = std::make_shared<GpuVoxelMap::Options>();
        if (node["resolutions"]) {
            auto resolutions_node = node["resolutions"];
            SLAM_CHECK_STREAM(resolutions_node.IsSequence(), "The node 'resolutions' in the yaml is not a sequence:\n" << node);
            map_options->resolutions.resize(0);
            for (auto child_node: resolutions_node) {
                SLAM_CHECK_STREAM(child_node.IsMap(), "The following child node is not a Map:\n" << child_node);
                MultipleResolutionVoxelMap::ResolutionParam param;
                FIND_OPTION(child_node, param, min_distance_between_points, double)
                FIND_OPTION(child_node, param, max_num_points, double)
                SLAM_CHECK_STREAM(child_node["resolution"], "Invalid Resolution Param in the yaml:\n" << child_node);
                param.resolution = child_node["resolution"].as<double>();
                map_options->resolutions.push_back(param);
            }
            SLAM_CHECK_STREAM(!map_options->resolutions.empty(), "The yaml does not define a valid set of resolutions for the map");
            SLAM_CHECK_STREAM(map_options->resolutions.size() <= CTGN_MAX_RESOLUTIONS, "too many resolutions for libctgn");
            std::sort(map_options->resolutions.begin(), map_options->resolutions.end(),
                      [](const auto &lhs, const auto &rhs) { return lhs.resolution < rhs.resolution; });
        }
        FIND_OPTION(node, (*map_options), max_frames_to_keep, int)
        FIND_OPTION(node, (*map_options), default_radius, double)
        FIND_OPTION(node, (*map_options), device, int)
        FIND_OPTION(node, (*map_options), device_updates, bool)
        FIND_OPTION(node, (*map_options), frame_pipeline, bool)
        FIND_OPTION(node, (*map_options), frame_shuffle_on_device, bool)
        return map_options;
    }

} // namespace ct_icp

#endif //CT_ICP_GPU_MAP_H
