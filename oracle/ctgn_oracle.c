/*
 * ctgn_oracle.c — CPU restatement of the reference's GN CT-ICP path. See ctgn_oracle.h for the status
 * of this file (test infrastructure only; parity pinned against oracle/_ref) and for what is restated from third parties.
 *
 * Every function cites the reference lines it follows (paths relative to the reference root).
 */
#include "ctgn_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ================================================================================================
 * Voxel map — include/ct_icp/map.h
 * ============================================================================================== */

typedef struct {
    int x, y, z;      /* slam::Voxel (include/SlamCore/types.h:65-86) */
    int count;
    double *pts;      /* count x 3, insertion order (map.h:480 iterates in this order) */
} orc_voxel;

typedef struct {
    orc_resolution param;
    orc_voxel *voxels;      /* dense list of live voxels */
    size_t num_voxels, cap_voxels;
    int64_t *table;         /* open addressing: index into voxels, -1 empty */
    size_t table_cap;       /* power of two */
    uint64_t num_points;
} orc_level;

struct orc_map {
    int num_levels;
    double default_radius;
    orc_level levels[ORC_MAX_RESOLUTIONS];
};

/* Eigen's association order for a 3-vector reduction (squaredNorm / norm / dot of Vector3d): the completely unrolled scalar redux
 * splits [0, 3) into [0, 1) and [1, 3) (Eigen/src/Core/Redux.h, redux_novec_unroller), i.e. c0 + (c1 + c2) -- NOT (c0 + c1) + c2.
 * With the reference's stock build flags (no -march, so no FMA) this is the rounding that decides `distance > radius`
 * (map.h:491-493), the heap comparisons (:494-500), the min-distance insert test (:279-286) and the eviction test (:313). */
static inline double sq_norm3(double dx, double dy, double dz) { return dx * dx + (dy * dy + dz * dz); }

static inline uint64_t orc_hash3(int x, int y, int z) {
    /* any hash works: bucket order is not observable through the queries.
     * (the reference's is x*73856093 + y*19349669 + z*83492791, types.h:610-623) */
    uint64_t h = (uint64_t) (uint32_t) x * 73856093ull + (uint64_t) (uint32_t) y * 19349669ull +
                 (uint64_t) (uint32_t) z * 83492791ull;
    h ^= h >> 29;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 32;
    return h;
}

static void level_rehash(orc_level *L, size_t new_cap) {
    free(L->table);
    L->table_cap = new_cap;
    L->table = (int64_t *) malloc(sizeof(int64_t) * new_cap);
    for (size_t i = 0; i < new_cap; ++i) L->table[i] = -1;
    for (size_t v = 0; v < L->num_voxels; ++v) {
        const orc_voxel *vx = &L->voxels[v];
        size_t s = (size_t) (orc_hash3(vx->x, vx->y, vx->z) & (new_cap - 1));
        while (L->table[s] >= 0) s = (s + 1) & (new_cap - 1);
        L->table[s] = (int64_t) v;
    }
}

static int64_t level_find(const orc_level *L, int x, int y, int z) {
    if (L->table_cap == 0) return -1;
    size_t s = (size_t) (orc_hash3(x, y, z) & (L->table_cap - 1));
    for (;;) {
        int64_t v = L->table[s];
        if (v < 0) return -1;
        const orc_voxel *vx = &L->voxels[v];
        if (vx->x == x && vx->y == y && vx->z == z) return v;
        s = (s + 1) & (L->table_cap - 1);
    }
}

static orc_voxel *level_add_voxel(orc_level *L, int x, int y, int z) {
    if ((L->num_voxels + 1) * 2 > L->table_cap) level_rehash(L, L->table_cap ? L->table_cap * 2 : 1024);
    if (L->num_voxels == L->cap_voxels) {
        L->cap_voxels = L->cap_voxels ? L->cap_voxels * 2 : 1024;
        L->voxels = (orc_voxel *) realloc(L->voxels, sizeof(orc_voxel) * L->cap_voxels);
    }
    orc_voxel *vx = &L->voxels[L->num_voxels];
    vx->x = x; vx->y = y; vx->z = z; vx->count = 0;
    vx->pts = (double *) malloc(sizeof(double) * 3 * (size_t) (L->param.max_num_points > 0 ? L->param.max_num_points : 1));
    size_t s = (size_t) (orc_hash3(x, y, z) & (L->table_cap - 1));
    while (L->table[s] >= 0) s = (s + 1) & (L->table_cap - 1);
    L->table[s] = (int64_t) L->num_voxels;
    L->num_voxels++;
    return vx;
}

/* Voxel::Coordinates — src/SlamCore/types.cxx:13-20: int(p / voxel_size), truncation toward zero. */
int orc_voxel_coord(double p, double voxel_size) { return (int) (p / voxel_size); }

orc_map *orc_map_create(const orc_resolution *res, int num_resolutions, double default_radius) {
    if (num_resolutions < 1 || num_resolutions > ORC_MAX_RESOLUTIONS) return NULL;
    orc_map *m = (orc_map *) calloc(1, sizeof(orc_map));
    m->num_levels = num_resolutions;
    m->default_radius = default_radius;
    for (int i = 0; i < num_resolutions; ++i) m->levels[i].param = res[i];
    return m;
}

void orc_map_clear(orc_map *m) {
    for (int l = 0; l < m->num_levels; ++l) {
        orc_level *L = &m->levels[l];
        for (size_t v = 0; v < L->num_voxels; ++v) free(L->voxels[v].pts);
        free(L->voxels); free(L->table);
        L->voxels = NULL; L->table = NULL;
        L->num_voxels = L->cap_voxels = 0; L->table_cap = 0; L->num_points = 0;
    }
}

void orc_map_destroy(orc_map *m) {
    if (!m) return;
    orc_map_clear(m);
    free(m);
}

/* InsertPointInVoxelMap — include/ct_icp/map.h:261-293.
 *   new voxel -> always insert; voxel with < max_num_points -> insert iff the min squared distance to
 *   the voxel's own points is STRICTLY greater than min_dist^2; full voxel -> drop. */
static int level_insert_point(orc_level *L, const double p[3]) {
    const double resolution = L->param.resolution, min_dist = L->param.min_distance_between_points;
    const int max_num_points = L->param.max_num_points;
    int vx = orc_voxel_coord(p[0], resolution), vy = orc_voxel_coord(p[1], resolution),
        vz = orc_voxel_coord(p[2], resolution);
    int64_t idx = level_find(L, vx, vy, vz);
    if (idx < 0) {
        orc_voxel *v = level_add_voxel(L, vx, vy, vz);      /* map.h:267-273 */
        v->pts[0] = p[0]; v->pts[1] = p[1]; v->pts[2] = p[2];
        v->count = 1;
        L->num_points++;
        return 1;
    }
    orc_voxel *v = &L->voxels[idx];
    if (v->count < max_num_points) {                        /* map.h:275-291 */
        double sq_dist_min = DBL_MAX;
        for (int i = 0; i < v->count; ++i) {
            double dx = v->pts[3 * i] - p[0], dy = v->pts[3 * i + 1] - p[1], dz = v->pts[3 * i + 2] - p[2];
            double sq = sq_norm3(dx, dy, dz);
            if (sq < sq_dist_min) sq_dist_min = sq;
        }
        if (sq_dist_min > min_dist * min_dist) {
            v->pts[3 * v->count] = p[0]; v->pts[3 * v->count + 1] = p[1]; v->pts[3 * v->count + 2] = p[2];
            v->count++;
            L->num_points++;
            return 1;
        }
    }
    return 0;
}

/* InsertPointCloud inner loop — map.h:196-206: every point goes to every resolution. */
void orc_map_insert(orc_map *m, const double *xyz, size_t n, uint8_t *inserted) {
    for (size_t i = 0; i < n; ++i) {
        int any = 0;
        for (int l = 0; l < m->num_levels; ++l) any |= level_insert_point(&m->levels[l], xyz + 3 * i);
        if (inserted) inserted[i] = (uint8_t) any;
    }
}

/* RemoveElementsFarFromLocation — map.h:305-322: a voxel goes iff its FIRST point is farther than
 * `distance` (norm, strict >) from `location`. (Empty voxels cannot exist: a voxel is born with one point.) */
void orc_map_remove_far(orc_map *m, const double location[3], double distance) {
    for (int l = 0; l < m->num_levels; ++l) {
        orc_level *L = &m->levels[l];
        size_t w = 0;
        for (size_t v = 0; v < L->num_voxels; ++v) {
            orc_voxel *vx = &L->voxels[v];
            double dx = vx->pts[0] - location[0], dy = vx->pts[1] - location[1], dz = vx->pts[2] - location[2];
            double d = sqrt(sq_norm3(dx, dy, dz));
            if (d > distance) {
                L->num_points -= (uint64_t) vx->count;
                free(vx->pts);
            } else {
                L->voxels[w++] = *vx;
            }
        }
        L->num_voxels = w;
        if (L->table_cap) level_rehash(L, L->table_cap);
    }
}

uint64_t orc_map_num_points(const orc_map *m) {
    uint64_t s = 0;
    for (int l = 0; l < m->num_levels; ++l) s += m->levels[l].num_points;
    return s;
}

uint64_t orc_map_num_voxels(const orc_map *m, int res_index) {
    if (res_index < 0 || res_index >= m->num_levels) return 0;
    return m->levels[res_index].num_voxels;
}

uint64_t orc_map_export(const orc_map *m, int res_index, double *out_xyz, uint64_t capacity_points) {
    if (res_index < 0 || res_index >= m->num_levels) return 0;
    const orc_level *L = &m->levels[res_index];
    uint64_t k = 0;
    for (size_t v = 0; v < L->num_voxels; ++v)
        for (int i = 0; i < L->voxels[v].count; ++i) {
            if (out_xyz && k < capacity_points) memcpy(out_xyz + 3 * k, L->voxels[v].pts + 3 * i, 3 * sizeof(double));
            ++k;
        }
    return k;
}

/* SearchParamsFromRadiusSearch — map.h:416-432: lower_bound with `lhs.resolution <= radius`, i.e. the first
 * resolution strictly greater than the radius, minus one, clamped at 0; sweep = ceil(radius/resolution). */
void orc_map_search_params(const orc_map *m, double radius, int *map_id, double *voxel_resolution,
                           int *voxel_neighborhood) {
    int it = 0;
    while (it < m->num_levels && m->levels[it].param.resolution <= radius) ++it;
    int idx = it - 1;
    if (idx < 0) idx = 0;
    double resolution = m->levels[idx].param.resolution;
    if (map_id) *map_id = idx;
    if (voxel_resolution) *voxel_resolution = resolution;
    if (voxel_neighborhood) *voxel_neighborhood = (int) ceil(radius / resolution);
}

/* ---- libstdc++ std::priority_queue<tuple<double,...>, vector, Comparator(a.d < b.d)> restated:
 *      push = push_back + std::push_heap, pop = std::pop_heap + pop_back (bits/stl_heap.h). ---- */
typedef struct { double d; double p[3]; uint32_t visit; } heap_item;

static inline int heap_less(const heap_item *a, const heap_item *b, int total_order) {
    if (a->d < b->d) return 1;
    if (total_order && a->d == b->d && a->visit < b->visit) return 1;
    return 0;
}

static void heap_push_hole(heap_item *first, long hole, long top, heap_item value, int total_order) {
    long parent = (hole - 1) / 2;
    while (hole > top && heap_less(&first[parent], &value, total_order)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void heap_push(heap_item *h, int *size, heap_item value, int total_order) {
    h[*size] = value;
    (*size)++;
    heap_push_hole(h, *size - 1, 0, h[*size - 1], total_order);
}

static void heap_adjust(heap_item *first, long hole, long len, heap_item value, int total_order) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (heap_less(&first[child], &first[child - 1], total_order)) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    heap_push_hole(first, hole, top, value, total_order);
}

static void heap_pop(heap_item *h, int *size, int total_order) {
    if (*size > 1) {
        heap_item value = h[*size - 1];
        h[*size - 1] = h[0];
        heap_adjust(h, 0, *size - 1, value, total_order);
    }
    (*size)--;
}

/* RadiusSearchInPlace — map.h:449-514 with sensor_location == nullptr (as ComputeNeighborhoodInPlace
 * passes it, map.h:527-530), so the normal-direction filter (:482-490) is inactive.
 * heap_mode 0: distances are sqrt'd norms and ties follow the libstdc++ heap exactly.
 * heap_mode 1: the candidate order is the total order (squared distance, visit index), the radius test
 *              stays `norm > radius`. Identical to mode 0 whenever no two candidates tie. */
int orc_map_radius_search(const orc_map *m, const double query[3], double radius, int max_num_neighbors,
                          int heap_mode, double *out_xyz) {
    if (radius <= 0) radius = m->default_radius;
    if (max_num_neighbors > ORC_MAX_NEIGHBORS) max_num_neighbors = ORC_MAX_NEIGHBORS;
    if (max_num_neighbors <= 0) return 0;
    int map_id, nb;
    double voxel_size;
    orc_map_search_params(m, radius, &map_id, &voxel_size, &nb);
    const orc_level *L = &m->levels[map_id];
    int kx = orc_voxel_coord(query[0], voxel_size), ky = orc_voxel_coord(query[1], voxel_size),
        kz = orc_voxel_coord(query[2], voxel_size);
    /* `short` sweep counters (map.h:470-472). Outside the int16 range the reference's loops do not
     * terminate; the oracle (and the product) define that case as "no neighbours". */
    if (kx - nb < -32768 || kx + nb + 1 > 32767 || ky - nb < -32768 || ky + nb + 1 > 32767 ||
        kz - nb < -32768 || kz + nb + 1 > 32767)
        return 0;
    heap_item heap[ORC_MAX_NEIGHBORS + 1];
    int hsize = 0;
    uint32_t visit = 0;
    for (short kxx = (short) (kx - nb); kxx < kx + nb + 1; ++kxx)
        for (short kyy = (short) (ky - nb); kyy < ky + nb + 1; ++kyy)
            for (short kzz = (short) (kz - nb); kzz < kz + nb + 1; ++kzz) {
                int64_t vi = level_find(L, kxx, kyy, kzz);
                if (vi < 0) continue;
                const orc_voxel *vb = &L->voxels[vi];
                for (int i = 0; i < vb->count; ++i, ++visit) {
                    double dx = vb->pts[3 * i] - query[0], dy = vb->pts[3 * i + 1] - query[1],
                           dz = vb->pts[3 * i + 2] - query[2];
                    double sq = sq_norm3(dx, dy, dz);
                    double distance = sqrt(sq);                       /* map.h:491 (.norm()) */
                    if (distance > radius) continue;                  /* map.h:492 */
                    heap_item it;
                    it.d = heap_mode ? sq : distance;
                    it.p[0] = vb->pts[3 * i]; it.p[1] = vb->pts[3 * i + 1]; it.p[2] = vb->pts[3 * i + 2];
                    it.visit = visit;
                    if (hsize == max_num_neighbors) {                 /* map.h:494-500 */
                        if (heap_less(&it, &heap[0], heap_mode)) {
                            heap_pop(heap, &hsize, heap_mode);
                            heap_push(heap, &hsize, it, heap_mode);
                        }
                    } else {
                        heap_push(heap, &hsize, it, heap_mode);
                    }
                }
            }
    int n = 0;
    while (hsize > 0) {                                               /* map.h:508-513: farthest first */
        if (out_xyz) { out_xyz[3 * n] = heap[0].p[0]; out_xyz[3 * n + 1] = heap[0].p[1]; out_xyz[3 * n + 2] = heap[0].p[2]; }
        ++n;
        heap_pop(heap, &hsize, heap_mode);
    }
    return n;
}

void orc_map_count(const orc_map *m, const double query[3], uint64_t *probed, uint64_t *hit, uint64_t *points) {
    int map_id, nb;
    double voxel_size;
    orc_map_search_params(m, m->default_radius, &map_id, &voxel_size, &nb);
    const orc_level *L = &m->levels[map_id];
    int kx = orc_voxel_coord(query[0], voxel_size), ky = orc_voxel_coord(query[1], voxel_size),
        kz = orc_voxel_coord(query[2], voxel_size);
    for (int x = kx - nb; x <= kx + nb; ++x)
        for (int y = ky - nb; y <= ky + nb; ++y)
            for (int z = kz - nb; z <= kz + nb; ++z) {
                (*probed)++;
                int64_t vi = level_find(L, x, y, z);
                if (vi >= 0) { (*hit)++; (*points) += (uint64_t) L->voxels[vi].count; }
            }
}

/* ================================================================================================
 * Geometry — Eigen 3 semantics restated. Quaternions are (x, y, z, w).
 * ============================================================================================== */

/* TPose::GetAlphaTimestamp — include/SlamCore/types.h:192-219. NB: returns 0 (not 1) above the max. */
double orc_alpha_timestamp(double t, double t_begin, double t_end) {
    double lo = t_begin < t_end ? t_begin : t_end;   /* std::min(dest_timestamp, other.dest_timestamp) */
    double hi = t_begin < t_end ? t_end : t_begin;
    if (lo > t) return 0.0;
    if (hi < t) return 0.0;
    if (lo == hi) return 1.0;
    return (t - lo) / (hi - lo);
}

void orc_quat_normalize(double q[4]) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* Eigen QuaternionBase::_transformVector: uv = 2 * (q_v x v); v + w*uv + q_v x uv. */
void orc_quat_rotate(const double q[4], const double v[3], double out[3]) {
    double uvx = q[1] * v[2] - q[2] * v[1], uvy = q[2] * v[0] - q[0] * v[2], uvz = q[0] * v[1] - q[1] * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    out[0] = v[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
    out[1] = v[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
    out[2] = v[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}

/* Eigen QuaternionBase::slerp. */
void orc_quat_slerp(const double a[4], const double b[4], double t, double out[4]) {
    const double one = 1.0 - DBL_EPSILON;
    double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    double absD = fabs(d), scale0, scale1;
    if (absD >= one) {
        scale0 = 1.0 - t; scale1 = t;
    } else {
        double theta = acos(absD), sinTheta = sin(theta);
        scale0 = sin((1.0 - t) * theta) / sinTheta;
        scale1 = sin(t * theta) / sinTheta;
    }
    if (d < 0) scale1 = -scale1;
    for (int i = 0; i < 4; ++i) out[i] = scale0 * a[i] + scale1 * b[i];
}

/* Eigen QuaternionBase::toRotationMatrix (row-major output). */
void orc_quat_to_matrix(const double q[4], double R[9]) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* Eigen quaternionbase_assign_impl<Matrix3> (trace / largest-diagonal branches). */
void orc_matrix_to_quat(const double R[9], double q[4]) {
#define M(i, j) R[3 * (i) + (j)]
    double t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (M(2, 1) - M(1, 2)) * t;
        q[1] = (M(0, 2) - M(2, 0)) * t;
        q[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (M(k, j) - M(j, k)) * t;
        q[j] = (M(j, i) + M(i, j)) * t;
        q[k] = (M(k, i) + M(i, k)) * t;
    }
#undef M
}

/* pose_begin.InterpolatePose(pose_end, t) * raw — types.h:453-470 (alpha via GetAlphaTimestamp),
 * TSE3::Interpolate :360-366 (slerp + lerp), TSE3::operator*(Tr) :353-357 (quat.normalized()*p + tr). */
void orc_transform_point(const double pose[14], const double tbe[2], double t, const double raw[3], double out[3]) {
    double alpha = orc_alpha_timestamp(t, tbe[0], tbe[1]);
    double q[4];
    orc_quat_slerp(pose, pose + 7, alpha, q);
    double tr[3];
    for (int i = 0; i < 3; ++i) tr[i] = (1.0 - alpha) * pose[4 + i] + alpha * pose[11 + i];
    orc_quat_normalize(q);
    double r[3];
    orc_quat_rotate(q, raw, r);
    out[0] = r[0] + tr[0]; out[1] = r[1] + tr[1]; out[2] = r[2] + tr[2];
}

/* Batch form: the full-scan undistortion loop of Odometry::DoRegister (src/ct_icp/odometry.cpp:461-486, the reference's
 * only OpenMP loop on this route). */
void orc_transform_points(const double pose[14], const double tbe[2], const double *t, const double *raw, size_t n,
                          double *out, int num_threads) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads > 1 ? num_threads : 1) schedule(static)
#endif
    for (long i = 0; i < (long) n; ++i) orc_transform_point(pose, tbe, t[i], raw + 3 * i, out + 3 * i);
}

/* Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations. Stands in for
 * Eigen::JacobiSVD<Matrix3d>(C, ComputeFullV) on a symmetric matrix (neighborhood.h:293): singular values
 * = |eigenvalues| sorted descending, V = eigenvectors (sign per column arbitrary; the caller orients). */
void orc_sym_eigen3(const double C[9], double evals[3], double V[9]) {
    double a[3][3], v[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { a[i][j] = 0.5 * (C[3 * i + j] + C[3 * j + i]); v[i][j] = (i == j); }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off == 0.0) break;
        double scale = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 + 1e-22 * scale) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                double app = a[p][p], aqq = a[q][q];
                a[p][p] = app - t * apq;
                a[q][q] = aqq + t * apq;
                a[p][q] = a[q][p] = 0.0;
                int r = 3 - p - q;
                double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
                for (int k = 0; k < 3; ++k) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int order[3] = {0, 1, 2};
    double mag[3] = {fabs(a[0][0]), fabs(a[1][1]), fabs(a[2][2])};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (mag[order[j]] < mag[order[j + 1]]) { int tmp = order[j]; order[j] = order[j + 1]; order[j + 1] = tmp; }
    for (int c = 0; c < 3; ++c) {
        evals[c] = a[order[c]][order[c]];
        for (int k = 0; k < 3; ++k) V[3 * k + c] = v[k][order[c]];
    }
}

/* Eigen::JacobiSVD<Matrix3d>(C, ComputeFullV) restated operation for operation (Eigen/src/SVD/JacobiSVD.h: two-sided Jacobi, sweeps
 * over (p, q) = (1,0), (2,0), (2,1), threshold 2 eps * max|diag|, real_2x2_jacobi_svd + JacobiRotation::makeJacobi from
 * Eigen/src/Jacobi/Jacobi.h, singular values |diag| * scale sorted in decreasing order with the columns of V). The same statement
 * as oracle/shims/mini_eigen.h (which oracle/_ref runs under the reference's own ComputeNeighborhoodInfo, neighborhood.h:285-316)
 * and as jacobi_svd3_exact of the product: on a rank-deficient covariance (collinear neighbours: a pole) the "normal" is decided by
 * these very roundings, so the three agree bit for bit only if they perform the same operations. sv descending, V row-major. */
void orc_jacobi_svd3(const double Cin[9], double sv[3], double V[9]) {
    double W[3][3], Vm[3][3];
    double scale = 0.0;
    for (int i = 0; i < 9; ++i) if (fabs(Cin[i]) > scale) scale = fabs(Cin[i]);
    if (scale == 0.0) scale = 1.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { W[i][j] = Cin[3 * i + j] / scale; Vm[i][j] = (i == j) ? 1.0 : 0.0; }
    const double precision = 2.0 * DBL_EPSILON, tiny = DBL_MIN;
    double max_diag = 0.0;
    for (int i = 0; i < 3; ++i) if (fabs(W[i][i]) > max_diag) max_diag = fabs(W[i][i]);
    int finished = 0;
    while (!finished) {
        finished = 1;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                double thr = precision * max_diag;
                if (thr < tiny) thr = tiny;
                if (!(fabs(W[p][q]) > thr || fabs(W[q][p]) > thr)) continue;
                finished = 0;
                /* real_2x2_jacobi_svd */
                const double m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
                double r1c, r1s;
                const double t = m00 + m11, d = m10 - m01;
                if (fabs(d) < tiny) { r1s = 0.0; r1c = 1.0; }
                else { const double u = t / d; const double tmp = sqrt(1.0 + u * u); r1s = 1.0 / tmp; r1c = u / tmp; }
                const double n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11, n11 = -r1s * m01 + r1c * m11;
                /* JacobiRotation::makeJacobi(n00, n01, n11) */
                double jc, js;
                const double deno = 2.0 * fabs(n01);
                if (deno < tiny) { jc = 1.0; js = 0.0; }
                else {
                    const double tau = (n00 - n11) / deno, w = sqrt(tau * tau + 1.0);
                    const double t2 = tau > 0.0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
                    const double sign_t = t2 > 0.0 ? 1.0 : -1.0;
                    const double n = 1.0 / sqrt(t2 * t2 + 1.0);
                    js = -sign_t * (n01 / fabs(n01)) * fabs(t2) * n;
                    jc = n;
                }
                /* j_left = rot1 * j_right^T */
                const double lc = r1c * jc - r1s * (-js), ls = r1c * (-js) + r1s * jc;
                if (!(lc == 1.0 && ls == 0.0))
                    for (int k = 0; k < 3; ++k) {
                        const double xi = W[p][k], yi = W[q][k];
                        W[p][k] = lc * xi + ls * yi;
                        W[q][k] = -ls * xi + lc * yi;
                    }
                /* applyOnTheRight(p, q, j_right): columns rotated by j_right^T = (jc, -js) */
                const double tc = jc, ts = -js;
                if (!(tc == 1.0 && ts == 0.0)) {
                    for (int k = 0; k < 3; ++k) {
                        const double xi = W[k][p], yi = W[k][q];
                        W[k][p] = tc * xi + ts * yi;
                        W[k][q] = -ts * xi + tc * yi;
                    }
                    for (int k = 0; k < 3; ++k) {
                        const double xi = Vm[k][p], yi = Vm[k][q];
                        Vm[k][p] = tc * xi + ts * yi;
                        Vm[k][q] = -ts * xi + tc * yi;
                    }
                }
                double md = fabs(W[p][p]) > fabs(W[q][q]) ? fabs(W[p][p]) : fabs(W[q][q]);
                if (md > max_diag) max_diag = md;
            }
    }
    for (int i = 0; i < 3; ++i) sv[i] = fabs(W[i][i]) * scale;
    for (int i = 0; i < 3; ++i) {
        int pos = 0;
        double mx = sv[i];
        for (int k = i + 1; k < 3; ++k) if (sv[k] > mx) { mx = sv[k]; pos = k - i; }
        if (mx == 0.0) break;
        if (pos) {
            pos += i;
            double ts = sv[i]; sv[i] = sv[pos]; sv[pos] = ts;
            for (int k = 0; k < 3; ++k) { double tv = Vm[k][pos]; Vm[k][pos] = Vm[k][i]; Vm[k][i] = tv; }
        }
    }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[3 * i + j] = Vm[i][j];
}

/* TNeighborhood::ComputeNeighborhood(A2D | NORMAL) — include/SlamCore/experimental/neighborhood.h:225-257
 * then ComputeNeighborhoodInfo :285-316. */
int orc_neighborhood(const double *pts, int n, double normal[3], double *a2d) {
    if (n < 5) return 0;                                     /* :227-230, MinNeighborhoodSize :184 */
    double bary[3] = {0, 0, 0}, cov[9] = {0};
    for (int i = 0; i < n; ++i) {                            /* :236-240 */
        const double *p = pts + 3 * i;
        for (int r = 0; r < 3; ++r) {
            bary[r] += p[r];
            for (int c = 0; c < 3; ++c) cov[3 * r + c] += p[r] * p[c];
        }
    }
    for (int r = 0; r < 3; ++r) bary[r] /= (double) n;       /* :241 */
    for (int k = 0; k < 9; ++k) cov[k] /= (double) n;        /* :242 */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) cov[3 * r + c] -= bary[r] * bary[c];   /* :243 */
    double ev[3], V[9];
    orc_jacobi_svd3(cov, ev, V);                             /* :293 Eigen::JacobiSVD<Mat3>(covariance, ComputeFullV) */
    normal[0] = V[2]; normal[1] = V[5]; normal[2] = V[8];    /* V.block<3,1>(0,2) :300-303 */
    double s0 = fabs(ev[0]), s1 = fabs(ev[1]), s2 = fabs(ev[2]);   /* singularValues().cwiseAbs() :304 */
    *a2d = (sqrt(s1) - sqrt(s2)) / sqrt(s0);                 /* :309-311 */
    return 1;
}

/* Eigen LDLT<Matrix<double,12,12>>::compute + solve restated (symmetric diagonal pivoting, unblocked
 * left-looking factorisation on the lower triangle, solve = P^T L^-T D^-1 L^-1 P b). */
void orc_ldlt_solve12(const double Ain[144], const double bin[12], double x[12]) {
    enum { N = 12 };
    double m[N][N], temp[N];
    int transp[N];
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) m[i][j] = Ain[N * i + j];
    for (int k = 0; k < N; ++k) {
        int big = k;
        double best = fabs(m[k][k]);
        for (int i = k + 1; i < N; ++i) if (fabs(m[i][i]) > best) { best = fabs(m[i][i]); big = i; }
        transp[k] = big;
        if (big != k) {
            /* symmetric swap of rows/cols k and big acting on the lower triangle */
            int s = N - big - 1;
            for (int j = 0; j < k; ++j) { double t = m[k][j]; m[k][j] = m[big][j]; m[big][j] = t; }
            for (int i = 0; i < s; ++i) { double t = m[big + 1 + i][k]; m[big + 1 + i][k] = m[big + 1 + i][big]; m[big + 1 + i][big] = t; }
            for (int i = k + 1; i < big; ++i) { double t = m[i][k]; m[i][k] = m[big][i]; m[big][i] = t; }
            { double t = m[k][k]; m[k][k] = m[big][big]; m[big][big] = t; }
        }
        int rs = N - k - 1;
        if (k > 0) {
            for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
            double acc = 0;
            for (int j = 0; j < k; ++j) acc += m[k][j] * temp[j];
            m[k][k] -= acc;
            for (int i = 0; i < rs; ++i) {
                double a2 = 0;
                for (int j = 0; j < k; ++j) a2 += m[k + 1 + i][j] * temp[j];
                m[k + 1 + i][k] -= a2;
            }
        }
        double akk = m[k][k];
        if (fabs(akk) > 0) for (int i = 0; i < rs; ++i) m[k + 1 + i][k] /= akk;
    }
    double y[N];
    for (int i = 0; i < N; ++i) y[i] = bin[i];
    for (int k = 0; k < N; ++k) if (transp[k] != k) { double t = y[k]; y[k] = y[transp[k]]; y[transp[k]] = t; }
    for (int i = 0; i < N; ++i) for (int j = 0; j < i; ++j) y[i] -= m[i][j] * y[j];
    const double tol = DBL_MIN;   /* Eigen: 1 / NumTraits<double>::highest() ~ 5.6e-309; DBL_MIN is the same gate in practice */
    for (int i = 0; i < N; ++i) y[i] = (fabs(m[i][i]) > tol) ? y[i] / m[i][i] : 0.0;
    for (int i = N - 1; i >= 0; --i) for (int j = i + 1; j < N; ++j) y[i] -= m[j][i] * y[j];
    for (int k = N - 1; k >= 0; --k) if (transp[k] != k) { double t = y[k]; y[k] = y[transp[k]]; y[transp[k]] = t; }
    for (int i = 0; i < N; ++i) x[i] = y[i];
}

/* ================================================================================================
 * The hot path — src/ct_icp/ct_icp.cpp:709-996
 * ============================================================================================== */

static inline double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* Body of the keypoint loop, ct_icp.cpp:753-857, for one keypoint. Returns 1 if it contributed. */
static int gn_keypoint(const orc_map *m, const double raw[3], const double world[3], double timestamp,
                       const double pose[14], const double tbe[2], const orc_options *o, int heap_mode,
                       double u[12], double *scalar_out, int32_t *n_nb, double *normal_out, double *a2d_out,
                       double *far_out) {
    double nb[3 * ORC_MAX_NEIGHBORS];
    /* :762 voxels_map.ComputeNeighborhood(pt_keypoint, max_number_neighbors) */
    int n = orc_map_radius_search(m, world, m->default_radius, o->max_number_neighbors, heap_mode, nb);
    if (n_nb) *n_nb = n;
    if (n < o->min_number_neighbors) return 0;                         /* :769 */
    double normal[3], a2d;
    if (!orc_neighborhood(nb, n, normal, &a2d)) return 0;              /* :778 (invalid below 5 points:
        the reference would then read an uninitialised normal; defined here as "skip") */
    const double *tb = pose + 4;
    if (normal[0] * (tb[0] - world[0]) + (normal[1] * (tb[1] - world[1]) + normal[2] * (tb[2] - world[2])) < 0) {   /* Eigen dot: c0 + (c1 + c2) */
        normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2];   /* :782-784 */
    }
    if (normal_out) { normal_out[0] = normal[0]; normal_out[1] = normal[1]; normal_out[2] = normal[2]; }
    if (a2d_out) *a2d_out = a2d;
    if (far_out) { far_out[0] = nb[0]; far_out[1] = nb[1]; far_out[2] = nb[2]; }
    double alpha = orc_alpha_timestamp(timestamp, tbe[0], tbe[1]);    /* :786 */
    double weight = a2d * a2d;                                          /* :787-788 */
    double cpn[3] = {weight * normal[0], weight * normal[1], weight * normal[2]};   /* :789 */
    const double *cp = nb;                                              /* :791 points[0] = farthest kept */
    double dist_to_plane = normal[0] * (world[0] - cp[0]) + normal[1] * (world[1] - cp[1]) +
                           normal[2] * (world[2] - cp[2]);              /* :793-795 */
    if (!(fabs(dist_to_plane) < o->max_dist_to_plane_ct_icp)) return 0; /* :803 */
    double scalar = cpn[0] * (world[0] - cp[0]) + cpn[1] * (world[1] - cp[1]) + cpn[2] * (world[2] - cp[2]); /* :805-807 */
    double ob[3], oe[3];
    orc_quat_rotate(pose, raw, ob);                                     /* :813-814 BeginQuat() * raw */
    orc_quat_rotate(pose + 7, raw, oe);                                 /* :815-816 */
    u[0] = (1 - alpha) * (ob[1] * cpn[2] - ob[2] * cpn[1]);             /* :818-826 */
    u[1] = (1 - alpha) * (ob[2] * cpn[0] - ob[0] * cpn[2]);
    u[2] = (1 - alpha) * (ob[0] * cpn[1] - ob[1] * cpn[0]);
    u[3] = (1 - alpha) * cpn[0]; u[4] = (1 - alpha) * cpn[1]; u[5] = (1 - alpha) * cpn[2];   /* :828-830 */
    u[6] = alpha * (oe[1] * cpn[2] - oe[2] * cpn[1]);                   /* :832-837 */
    u[7] = alpha * (oe[2] * cpn[0] - oe[0] * cpn[2]);
    u[8] = alpha * (oe[0] * cpn[1] - oe[1] * cpn[0]);
    u[9] = alpha * cpn[0]; u[10] = alpha * cpn[1]; u[11] = alpha * cpn[2];                   /* :839-841 */
    *scalar_out = scalar;
    return 1;
}

void orc_gn_accumulate(const orc_map *m, const double *raw_xyz, const double *world_xyz, const double *t,
                       size_t n, const double pose[14], const double tbe[2], const orc_options *o,
                       int heap_mode, int num_threads, double A[144], double b[12], int *n_used,
                       int32_t *n_neighbors, double *normal, double *a2d, double *farthest, uint8_t *used) {
    memset(A, 0, sizeof(double) * 144);                                /* :746-747 */
    memset(b, 0, sizeof(double) * 12);
    *n_used = 0;                                                       /* :749 */
#ifdef _OPENMP
    if (num_threads > 1) {
        int nt = num_threads;
        double *Ap = (double *) calloc((size_t) nt * 157, sizeof(double));
#pragma omp parallel num_threads(nt)
        {
            int tid = omp_get_thread_num();
            double *At = Ap + (size_t) tid * 157, *bt = At + 144;
            int cnt = 0;
#pragma omp for schedule(static)
            for (long pid = 0; pid < (long) n; ++pid) {
                double u[12], scalar;
                int ok = gn_keypoint(m, raw_xyz + 3 * pid, world_xyz + 3 * pid, t[pid], pose, tbe, o, heap_mode, u,
                                     &scalar, n_neighbors ? n_neighbors + pid : NULL, normal ? normal + 3 * pid : NULL,
                                     a2d ? a2d + pid : NULL, farthest ? farthest + 3 * pid : NULL);
                if (used) used[pid] = (uint8_t) ok;
                if (!ok) continue;
                cnt++;
                for (int i = 0; i < 12; ++i) {
                    for (int j = 0; j < 12; ++j) At[12 * i + j] += u[i] * u[j];
                    bt[i] -= u[i] * scalar;
                }
            }
            At[156] = (double) cnt;
        }
        for (int tid = 0; tid < nt; ++tid) {     /* ordered final sum */
            const double *At = Ap + (size_t) tid * 157;
            for (int k = 0; k < 144; ++k) A[k] += At[k];
            for (int k = 0; k < 12; ++k) b[k] += At[144 + k];
            *n_used += (int) At[156];
        }
        free(Ap);
        return;
    }
#else
    (void) num_threads;
#endif
    for (size_t pid = 0; pid < n; ++pid) {                             /* :753 — serial in the reference */
        double u[12], scalar;
        int ok = gn_keypoint(m, raw_xyz + 3 * pid, world_xyz + 3 * pid, t[pid], pose, tbe, o, heap_mode, u, &scalar,
                             n_neighbors ? n_neighbors + pid : NULL, normal ? normal + 3 * pid : NULL,
                             a2d ? a2d + pid : NULL, farthest ? farthest + 3 * pid : NULL);
        if (used) used[pid] = (uint8_t) ok;
        if (!ok) continue;
        (*n_used)++;                                                   /* :810 */
        for (int i = 0; i < 12; ++i) {                                 /* :845-850 */
            for (int j = 0; j < 12; ++j) A[12 * i + j] = A[12 * i + j] + u[i] * u[j];
            b[i] = b[i] - u[i] * scalar;
        }
    }
}

/* Euler increment exactly as spelled at ct_icp.cpp:919-932 (= Rz(gamma) Ry(beta) Rx(alpha)), row-major. */
static void euler_rotation(double al, double be, double ga, double R[9]) {
    R[0] = cos(ga) * cos(be);
    R[1] = -sin(ga) * cos(al) + cos(ga) * sin(be) * sin(al);
    R[2] = sin(ga) * sin(al) + cos(ga) * sin(be) * cos(al);
    R[3] = sin(ga) * cos(be);
    R[4] = cos(ga) * cos(al) + sin(ga) * sin(be) * sin(al);
    R[5] = -cos(ga) * sin(al) + sin(ga) * sin(be) * cos(al);
    R[6] = -sin(be);
    R[7] = cos(be) * sin(al);
    R[8] = cos(be) * cos(al);
}

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

double orc_gn_solve_update(double A[144], double b[12], int n_used, const orc_motion_prior *prior,
                           double pose[14], double x_out[12]) {
    for (int i = 0; i < 12; ++i) {                                     /* :877-882 */
        for (int j = 0; j < 12; ++j) A[12 * i + j] = A[12 * i + j] / n_used;
        b[i] = b[i] / n_used;
    }
    if (prior) {                                                       /* :885-910 */
        const double AC = prior->beta_location_consistency, AE = prior->beta_constant_velocity;
        for (int k = 0; k < 3; ++k) {
            double diff_traj = pose[4 + k] - pose[11 + k];             /* BeginTr - EndTr :892 */
            A[13 * (3 + k)] += AC;
            b[3 + k] -= AC * diff_traj;
            double diff_ego = pose[11 + k] - pose[4 + k] - prior->previous_end_tr[k] + prior->previous_begin_tr[k];  /* :900-902 */
            A[13 * (9 + k)] += AE;
            b[9 + k] -= AE * diff_ego;
        }
    }
    double x[12];
    orc_ldlt_solve12(A, b, x);                                         /* :914 */
    double Rb[9], Re[9], Q[9], P[9];
    euler_rotation(x[0], x[1], x[2], Rb);                              /* :916-932 */
    euler_rotation(x[6], x[7], x[8], Re);                              /* :935-947 */
    orc_quat_to_matrix(pose, Q);
    mat3_mul(Rb, Q, P);
    orc_matrix_to_quat(P, pose);                                       /* :950-951 */
    for (int k = 0; k < 3; ++k) pose[4 + k] += x[3 + k];               /* :952 */
    orc_quat_to_matrix(pose + 7, Q);
    mat3_mul(Re, Q, P);
    orc_matrix_to_quat(P, pose + 7);                                   /* :953-954 */
    for (int k = 0; k < 3; ++k) pose[11 + k] += x[9 + k];              /* :955 */
    orc_quat_normalize(pose);                                          /* :961-962 */
    orc_quat_normalize(pose + 7);
    double nrm = 0;
    for (int i = 0; i < 12; ++i) { nrm += x[i] * x[i]; if (x_out) x_out[i] = x[i]; }
    return sqrt(nrm);
}

int orc_register_gn(const orc_map *m, const double *raw_xyz, double *world_xyz, const double *t, size_t n,
                    double pose[14], const double tbe[2], const orc_options *o, const orc_motion_prior *prior,
                    int heap_mode, int num_threads, orc_summary *summary) {
    memset(summary, 0, sizeof(*summary));
    /* InterpolatePose CHECKs dest_timestamp <= t <= other.dest_timestamp (types.h:456) at :965 */
    for (size_t i = 0; i < n; ++i)
        if (!(tbe[0] <= t[i] && t[i] <= tbe[1])) return -5;
    orc_quat_normalize(pose);                                          /* :716-717 */
    orc_quat_normalize(pose + 7);
    double A[144], b[12];
    int n_used = 0;
    int iter = 0;
    for (; iter < o->num_iters_icp; ++iter) {                          /* :745 */
        double t0 = now_sec();
        orc_gn_accumulate(m, raw_xyz, world_xyz, t, n, pose, tbe, o, heap_mode, num_threads, A, b, &n_used,
                          NULL, NULL, NULL, NULL, NULL);
        double t1 = now_sec();
        summary->t_neighbors += t1 - t0;
        if (n_used < 100) {                                            /* :860-871 */
            snprintf(summary->error_log, sizeof(summary->error_log),
                     "[CT_ICP]Error : not enough keypoints selected in ct-icp !\n[CT_ICP]Number_of_residuals : %d\n",
                     n_used);
            if (o->debug_print) fputs(summary->error_log, stdout);
            summary->success = 0;
            summary->num_residuals_used = n_used;
            summary->num_iters = iter;
            return 0;
        }
        double x[12];
        double nrm = orc_gn_solve_update(A, b, n_used, prior, pose, x);     /* :877-962 */
        double t2 = now_sec();
        summary->t_solve += t2 - t1;
        if (num_threads > 1) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads) schedule(static)
#endif
            for (long pid = 0; pid < (long) n; ++pid)
                orc_transform_point(pose, tbe, t[pid], raw_xyz + 3 * pid, world_xyz + 3 * pid);
        } else {
            for (size_t pid = 0; pid < n; ++pid)                           /* :964-966 */
                orc_transform_point(pose, tbe, t[pid], raw_xyz + 3 * pid, world_xyz + 3 * pid);
        }
        summary->t_update += now_sec() - t2;
        summary->last_step_norm = nrm;
        if (nrm < o->threshold_orientation_norm) { ++iter; break; }   /* :978-980 */
    }
    summary->success = 1;                                              /* :992-993 */
    summary->num_residuals_used = n_used;
    summary->num_iters = iter;
    return 0;
}

/* sub_sample_frame — ct_icp.cpp:65-83: voxel = static_cast<short>(raw / size) per axis, first point per
 * voxel wins. Output in first-insertion order. */
size_t orc_grid_sampling(const double *raw_xyz, size_t n, double voxel_size, uint32_t *out_indices) {
    size_t cap = 1024;
    while (cap < 4 * n + 16) cap <<= 1;
    int64_t *table = (int64_t *) malloc(sizeof(int64_t) * cap);
    for (size_t i = 0; i < cap; ++i) table[i] = -1;
    size_t kept = 0;
    for (size_t i = 0; i < n; ++i) {
        short vx = (short) (raw_xyz[3 * i] / voxel_size), vy = (short) (raw_xyz[3 * i + 1] / voxel_size),
              vz = (short) (raw_xyz[3 * i + 2] / voxel_size);
        size_t s = (size_t) (orc_hash3(vx, vy, vz) & (cap - 1));
        int found = 0;
        while (table[s] >= 0) {
            size_t j = (size_t) table[s];
            short wx = (short) (raw_xyz[3 * j] / voxel_size), wy = (short) (raw_xyz[3 * j + 1] / voxel_size),
                  wz = (short) (raw_xyz[3 * j + 2] / voxel_size);
            if (wx == vx && wy == vy && wz == vz) { found = 1; break; }
            s = (s + 1) & (cap - 1);
        }
        if (!found) { table[s] = (int64_t) i; out_indices[kept++] = (uint32_t) i; }
    }
    free(table);
    return kept;
}

/* AdaptiveSamplePointsInGrid — include/ct_icp/algorithm/sampling.h:55-110 (the keypoint sampling of the NCLT profile,
 * odometry.cpp:539-545). Per point: range d = |p|; band = (first list entry with distance >= d) - 1, taken only when
 * distance[0] <= d < distance[last] (:69-76); voxel = int(p / voxel_size[band]) per axis (types.cxx:13-20); a voxel keeps
 * its first num_points_per_voxel indices (:80-85). d == distance[0] exactly indexes entry -1 in the reference (undefined
 * behaviour): such a point is dropped here. Emission (:93-108): band by band, voxel by voxel, stopping once MORE than
 * max_num_points indices have been written (the reference tests `size() > max`, so max + 1 survive). The reference walks
 * each band's std::unordered_map in its unspecified iteration order; the order fixed here is ascending (z, y, x) voxel
 * coordinate inside a band, indices ascending inside a voxel. Returns the number of indices written, or (size_t) -1 on an
 * invalid band list (not ascending, or a used band with a non-positive voxel size). */
typedef struct { int band, x, y, z; uint32_t first_slot, count; } orc_as_voxel;

static int orc_as_cmp(const void *a, const void *b) {
    const orc_as_voxel *u = (const orc_as_voxel *) a, *v = (const orc_as_voxel *) b;
    if (u->band != v->band) return u->band < v->band ? -1 : 1;
    if (u->z != v->z) return u->z < v->z ? -1 : 1;
    if (u->y != v->y) return u->y < v->y ? -1 : 1;
    if (u->x != v->x) return u->x < v->x ? -1 : 1;
    return 0;
}

size_t orc_adaptive_sampling(const double *raw_xyz, size_t n, int num_points_per_voxel, int max_num_points, int num_bands,
                             const double *distance, const double *voxel_size, uint32_t *out_indices) {
    if (num_bands < 2 || num_points_per_voxel < 1) return (size_t) -1;
    for (int j = 0; j + 1 < num_bands; ++j)
        if (!(distance[j] < distance[j + 1]) || !(voxel_size[j] > 0)) return (size_t) -1;
    const size_t k = (size_t) num_points_per_voxel;
    size_t cap = 1024;
    while (cap < 4 * n + 16) cap <<= 1;
    int64_t *table = (int64_t *) malloc(sizeof(int64_t) * cap);          /* slot -> voxel number */
    orc_as_voxel *vox = (orc_as_voxel *) malloc(sizeof(orc_as_voxel) * (n + 1));
    uint32_t *kept = (uint32_t *) malloc(sizeof(uint32_t) * (n + 1) * k);  /* voxel v keeps kept[v*k .. v*k+count) */
    for (size_t i = 0; i < cap; ++i) table[i] = -1;
    size_t nvox = 0;
    for (size_t i = 0; i < n; ++i) {
        const double x = raw_xyz[3 * i], y = raw_xyz[3 * i + 1], z = raw_xyz[3 * i + 2];
        const double d = sqrt(x * x + y * y + z * z);
        int lw = 0;                                                      /* std::lower_bound: first entry with distance >= d */
        while (lw < num_bands && distance[lw] < d) ++lw;
        if (!(d >= distance[0] && d < distance[num_bands - 1])) continue;
        const int band = lw - 1;
        if (band < 0) continue;                                          /* d == distance[0]: UB in the reference, dropped */
        const double sz = voxel_size[band];
        const int vx = (int) (x / sz), vy = (int) (y / sz), vz = (int) (z / sz);
        size_t s = (size_t) ((orc_hash3(vx, vy, vz) + 0x9E3779B97F4A7C15ull * (uint64_t) (band + 1)) & (cap - 1));
        int64_t v = -1;
        while (table[s] >= 0) {
            const orc_as_voxel *c = &vox[table[s]];
            if (c->band == band && c->x == vx && c->y == vy && c->z == vz) { v = table[s]; break; }
            s = (s + 1) & (cap - 1);
        }
        if (v < 0) {
            v = (int64_t) nvox++;
            table[s] = v;
            vox[v].band = band; vox[v].x = vx; vox[v].y = vy; vox[v].z = vz; vox[v].first_slot = (uint32_t) v; vox[v].count = 0;
        }
        if (vox[v].count < k) kept[(size_t) v * k + vox[v].count++] = (uint32_t) i;
    }
    qsort(vox, nvox, sizeof(orc_as_voxel), orc_as_cmp);
    const size_t limit = max_num_points > 0 ? (size_t) max_num_points : (size_t) 0x7fffffff;   /* kMaxNumPoints, :59 */
    size_t out = 0;
    for (size_t v = 0; v < nvox && out <= limit; ++v)
        for (uint32_t j = 0; j < vox[v].count && out <= limit; ++j)
            out_indices[out++] = kept[(size_t) vox[v].first_slot * k + j];
    free(table); free(vox); free(kept);
    return out;
}
