/*
 * ref_shaped.cpp — the SAME GN accumulation as ctgn_oracle.c, but on data structures shaped like the reference's:
 * a node-based hash map Voxel -> VoxelBlock{std::vector<80-byte point record>}, a std::priority_queue of
 * (distance, xyz, voxel) tuples, an 80-byte record copied out per candidate and one heap-allocated neighbour vector per
 * keypoint (reference include/ct_icp/map.h:326-339,449-514; slam::Neighborhood). TEST INFRASTRUCTURE / CPU BASELINE ONLY.
 *
 * Why: the flat-hash oracle is faster than the reference's implementation, which flatters the CPU side of the GPU / CPU
 * ratio. SURVEY.md section 8d asks for this variant to be timed "for honesty". std::unordered_map stands in for
 * tsl::robin_map (absent from this image); everything after the neighbour search calls the oracle's own functions, so the
 * packed system is identical to orc_gn_accumulate(heap_mode = 0) and tests/test_oracle_properties.py checks that.
 */
#include "ctgn_oracle.h"

#include <cmath>
#include <cstring>
#include <queue>
#include <tuple>
#include <unordered_map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct Voxel {
    int x, y, z;
    bool operator==(const Voxel &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct VoxelHash {      // std::hash<slam::Voxel>, include/SlamCore/types.h:610-623
    size_t operator()(const Voxel &v) const {
        return (size_t) v.x * 73856093u + (size_t) v.y * 19349669u + (size_t) v.z * 83492791u;
    }
};
struct Vec3 { double x, y, z; };
struct PointType {      // MultipleResolutionVoxelMap::PointType, map.h:326-339 (xyz, normal, flags, timestamp, ids)
    Vec3 xyz;
    Vec3 normal;
    bool is_normal_computed = false, is_normal_oriented = false;
    double timestamp = 0.0;
    uint32_t frame_id = 0, point_id = 0;
};
struct VoxelBlock { std::vector<PointType> points; };
using Map = std::unordered_map<Voxel, VoxelBlock, VoxelHash>;
using pq_item = std::tuple<double, Vec3, Voxel>;
struct Cmp { bool operator()(const pq_item &a, const pq_item &b) const { return std::get<0>(a) < std::get<0>(b); } };

struct RefMap {
    Map map;
    double resolution;
};

std::vector<Vec3> radius_search(const RefMap &m, const double q[3], double radius, int nb, int max_num_neighbors) {
    std::vector<Vec3> out;
    out.reserve(max_num_neighbors);
    const int kx = orc_voxel_coord(q[0], m.resolution), ky = orc_voxel_coord(q[1], m.resolution), kz = orc_voxel_coord(q[2], m.resolution);
    std::priority_queue<pq_item, std::vector<pq_item>, Cmp> pq;
    PointType neighbor;
    Voxel voxel{0, 0, 0};
    for (short kxx = (short) (kx - nb); kxx < kx + nb + 1; ++kxx)
        for (short kyy = (short) (ky - nb); kyy < ky + nb + 1; ++kyy)
            for (short kzz = (short) (kz - nb); kzz < kz + nb + 1; ++kzz) {
                voxel.x = kxx; voxel.y = kyy; voxel.z = kzz;
                auto it = m.map.find(voxel);
                if (it == m.map.end()) continue;
                const VoxelBlock &blk = it->second;
                for (size_t i = 0; i < blk.points.size(); ++i) {
                    neighbor = blk.points[i];                                   // 80-byte copy, as map.h:481
                    const double dx = neighbor.xyz.x - q[0], dy = neighbor.xyz.y - q[1], dz = neighbor.xyz.z - q[2];
                    const double distance = std::sqrt(dx * dx + (dy * dy + dz * dz));
                    if (distance > radius) continue;
                    if ((int) pq.size() == max_num_neighbors) {
                        if (distance < std::get<0>(pq.top())) {
                            pq.pop();
                            pq.emplace(distance, neighbor.xyz, voxel);
                        }
                    } else {
                        pq.emplace(distance, neighbor.xyz, voxel);
                    }
                }
            }
    while (!pq.empty()) {
        out.push_back(std::get<1>(pq.top()));
        pq.pop();
    }
    return out;
}

// ct_icp.cpp:753-857 for one keypoint, on the vector the search returned
int keypoint(const RefMap &m, const double raw[3], const double world[3], double timestamp, const double pose[14],
             const double tbe[2], const orc_options *o, double radius, int nb, double u[12], double *scalar_out) {
    std::vector<Vec3> nbv = radius_search(m, world, radius, nb, o->max_number_neighbors);
    const int n = (int) nbv.size();
    if (n < o->min_number_neighbors) return 0;
    double normal[3], a2d;
    if (!orc_neighborhood(&nbv[0].x, n, normal, &a2d)) return 0;
    const double *tb = pose + 4;
    if (normal[0] * (tb[0] - world[0]) + normal[1] * (tb[1] - world[1]) + normal[2] * (tb[2] - world[2]) < 0) {
        normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2];
    }
    const double alpha = orc_alpha_timestamp(timestamp, tbe[0], tbe[1]);
    const double weight = a2d * a2d;
    const double cpn[3] = {weight * normal[0], weight * normal[1], weight * normal[2]};
    const double *cp = &nbv[0].x;
    const double dist = normal[0] * (world[0] - cp[0]) + normal[1] * (world[1] - cp[1]) + normal[2] * (world[2] - cp[2]);
    if (!(std::fabs(dist) < o->max_dist_to_plane_ct_icp)) return 0;
    const double scalar = cpn[0] * (world[0] - cp[0]) + cpn[1] * (world[1] - cp[1]) + cpn[2] * (world[2] - cp[2]);
    double ob[3], oe[3];
    orc_quat_rotate(pose, raw, ob);
    orc_quat_rotate(pose + 7, raw, oe);
    u[0] = (1 - alpha) * (ob[1] * cpn[2] - ob[2] * cpn[1]);
    u[1] = (1 - alpha) * (ob[2] * cpn[0] - ob[0] * cpn[2]);
    u[2] = (1 - alpha) * (ob[0] * cpn[1] - ob[1] * cpn[0]);
    u[3] = (1 - alpha) * cpn[0]; u[4] = (1 - alpha) * cpn[1]; u[5] = (1 - alpha) * cpn[2];
    u[6] = alpha * (oe[1] * cpn[2] - oe[2] * cpn[1]);
    u[7] = alpha * (oe[2] * cpn[0] - oe[0] * cpn[2]);
    u[8] = alpha * (oe[0] * cpn[1] - oe[1] * cpn[0]);
    u[9] = alpha * cpn[0]; u[10] = alpha * cpn[1]; u[11] = alpha * cpn[2];
    *scalar_out = scalar;
    return 1;
}

}  // namespace

extern "C" {

/* Buckets the given points (one resolution level as exported by orc_map_export, i.e. already filtered by the insert rule)
 * in the given order. */
void *orc_refshaped_create(const double *xyz, size_t n, double resolution) {
    RefMap *m = new RefMap();
    m->resolution = resolution;
    for (size_t i = 0; i < n; ++i) {
        Voxel v{orc_voxel_coord(xyz[3 * i], resolution), orc_voxel_coord(xyz[3 * i + 1], resolution),
                orc_voxel_coord(xyz[3 * i + 2], resolution)};
        PointType p;
        p.xyz = Vec3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        p.normal = Vec3{0, 0, 1};
        p.point_id = (uint32_t) i;
        m->map[v].points.push_back(p);
    }
    return m;
}

void orc_refshaped_destroy(void *h) { delete static_cast<RefMap *>(h); }

void orc_refshaped_gn_accumulate(const void *h, const double *raw_xyz, const double *world_xyz, const double *t, size_t n,
                                 const double pose[14], const double tbe[2], const orc_options *o, double radius,
                                 int voxel_neighborhood, int num_threads, double A[144], double b[12], int *n_used) {
    const RefMap &m = *static_cast<const RefMap *>(h);
    std::memset(A, 0, sizeof(double) * 144);
    std::memset(b, 0, sizeof(double) * 12);
    *n_used = 0;
#ifdef _OPENMP
    if (num_threads > 1) {
        std::vector<double> Ap((size_t) num_threads * 157, 0.0);
#pragma omp parallel num_threads(num_threads)
        {
            double *At = Ap.data() + (size_t) omp_get_thread_num() * 157, *bt = At + 144;
            int cnt = 0;
#pragma omp for schedule(static)
            for (long pid = 0; pid < (long) n; ++pid) {
                double u[12], scalar;
                if (!keypoint(m, raw_xyz + 3 * pid, world_xyz + 3 * pid, t[pid], pose, tbe, o, radius, voxel_neighborhood, u, &scalar))
                    continue;
                cnt++;
                for (int i = 0; i < 12; ++i) {
                    for (int j = 0; j < 12; ++j) At[12 * i + j] += u[i] * u[j];
                    bt[i] -= u[i] * scalar;
                }
            }
            At[156] = (double) cnt;
        }
        for (int tid = 0; tid < num_threads; ++tid) {
            const double *At = Ap.data() + (size_t) tid * 157;
            for (int k = 0; k < 144; ++k) A[k] += At[k];
            for (int k = 0; k < 12; ++k) b[k] += At[144 + k];
            *n_used += (int) At[156];
        }
        return;
    }
#else
    (void) num_threads;
#endif
    for (size_t pid = 0; pid < n; ++pid) {
        double u[12], scalar;
        if (!keypoint(m, raw_xyz + 3 * pid, world_xyz + 3 * pid, t[pid], pose, tbe, o, radius, voxel_neighborhood, u, &scalar))
            continue;
        (*n_used)++;
        for (int i = 0; i < 12; ++i) {
            for (int j = 0; j < 12; ++j) A[12 * i + j] = A[12 * i + j] + u[i] * u[j];
            b[i] = b[i] - u[i] * scalar;
        }
    }
}

}  // extern "C"
