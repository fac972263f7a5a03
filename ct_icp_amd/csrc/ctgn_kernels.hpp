// ctgn_kernels.hpp — gfx950 (CDNA4, wave64) kernels of the Gauss–Newton CT-ICP path.
//
// One GN iteration of ct_icp::CT_ICP_Registration::DoRegisterGaussNewton (reference
// src/ct_icp/ct_icp.cpp:745-981) is three launches on one stream, with no host synchronisation:
//
//   k_accumulate_rows  per keypoint: [re-transform with the current pose, :964-966] -> voxel-hash neighbour
//                      search (include/ct_icp/map.h:449-514) -> neighbour record (k_accumulate_lane: search + the next step fused)
//   k_residual_reduce  mean/covariance -> normal + a2D (include/SlamCore/experimental/neighborhood.h:225-257,285-316) -> gates,
//                      residual, 12-vector u (:782-841) -> per-block packed sum of u u^T | -u r | count (:843-850)
//   k_reduce_solve   fixed-order sum of the per-block partials -> normalise, motion prior, LDL^T, pose
//                    update, stop test (:860-962, :978-980); writes the stop flag the next launches read.
//
// Two accumulate kernels share everything except the search:
//   k_accumulate_rows  (default) 16 lanes cooperate on one keypoint, 4 keypoints per wave in flight:
//                      the 27/125 hash probes of a keypoint are issued by the 16 lanes in parallel, a voxel's
//                      point block (an array of 24-byte points) is read in chunks of 16 points, candidates are compacted
//                      into a per-row LDS list with ballot/popcount, the k nearest are selected with DPP row operations;
//                      the kept points' byte offsets go to a per-keypoint record that k_residual_reduce consumes.
//   k_accumulate_lane  one lane per keypoint, sequential insertion list in LDS — the simple restatement used
//                      to cross-check the row kernel on the GPU (ctgn_set_variant(h, 1)).
//
// The search is gather + selection (no matrix pipe, as BASELINE.json's north_star says); the one dense contraction of the path — the
// 13 x 13 product U^T U of a wave's 64 residual records in k_residual_reduce — runs on v_mfma_f64_16x16x4_f64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "ctgn_map.hpp"
#include "ctgn_math.hpp"

namespace ctgn {

constexpr int SYS_N = 96;            // packed system: 78 upper-tri JtJ | 12 Jtr | count | pad
constexpr int SYS_USED = 91;
constexpr int KMAX = 32;             // CTGN_MAX_NEIGHBORS
constexpr int MAX_PARTIAL_BLOCKS = 2048;
constexpr int SEL_STRIDE = 36;       // 1 + KMAX, rounded to a 16-byte multiple

struct MapView {
    const Slot *slots;
    const double *blocks;
    uint32_t mask;
    int blk;                 // points per block (max_num_points)
    int nb;                  // sweep half-width (voxel_neighborhood)
    double resolution;
    double r2thr;            // radius_sq_threshold(radius)
};

struct KpView {
    const double *rx, *ry, *rz, *t;
    double *wx, *wy, *wz;
    uint32_t *sel;           // [n][SEL_STRIDE] per-keypoint hand-over of k_accumulate_rows: word 0 = n | pool size << 8 | TIE_FLAG, then the byte
                             // offsets of the pool's points nearest first — the first n are the neighbours
    uint32_t *cnt;           // [n] the record's count again, dense: the residual kernel looks here first and only fetches the records it will use
    float *kth;              // [n][2] left by the last search / pool check of the keypoint (k_accumulate_rows, phase V):
                             //   [0] radius round its position then inside which its pool is complete (rounded down; 0: no pool) — without
                             //       pools (k at the record's capacity): the k-th neighbour's distance (rounded up; 0: fewer than k)
                             //   [1] distance of its k-th neighbour then, or the search radius if it had fewer than k (rounded up)
    int kth_valid;           // 1: kth[] and the world points come from the previous search of THIS solve on the same map (set per launch)
    int pools;               // 1: bounded searches keep pools and later iterations check them first (rows_tiles, phase V); 0: every iteration searches
    int n;
    const uint32_t *order;   // k_accumulate_rows works on keypoint order[s] at position s (positions sorted by home voxel); nullptr = identity
    int chunk;               // rounds of a tile that take consecutive positions (>= 1)
    int xcd_split;           // 1: the blocks of XCD x (blockIdx % 8) work inside the x-th eighth of the tiles (needs gridDim >= 8)
    unsigned long long *clk_iter_start;   // &GnState::clk_iter_start (the search kernels see the state read-only); nullptr = do not stamp
    // split launches (k_pool_check in front of k_accumulate_rows, from the third search of a solve on):
    uint32_t *fail_list;     // k_pool_check appends the positions whose pool did not certify their neighbours ...
    int *fail_count;         // ... and counts them here (zero when the kernel starts); fail_count_next is zeroed for the next launch
    int *fail_count_next;
    const int *n_dev;        // k_accumulate_rows: the number of positions to work on lives on the device (min(*n_dev, n)); nullptr = n
    float guess2;            // > 0: a search that has no carried-over bound (the first of a solve) starts from this squared distance instead of the
                             // radius; a keypoint it leaves with fewer than k candidates is searched again on the radius (rows_tiles, pass 1)
    int resume;              // k_accumulate_rows: 1 = the positions come from `order` (= a fail list), their world points are current (the
                             // check kernel wrote them) and kth[1] is the bound to search within: no transform, nothing moved
};

// working copy of the keypoint block in position order (ctgn_api.hip, order_keypoints): dst[a][pos] = src[a][order[pos]]
__global__ void k_kp_permute(const double *src, size_t stride, const uint32_t *order, int n, double *dst, size_t dst_stride) {
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += gridDim.x * blockDim.x) {
        const size_t i = order[pos];
#pragma unroll
        for (int a = 0; a < 7; ++a) dst[a * dst_stride + pos] = src[a * stride + i];
    }
}

struct GnParams {
    int min_nb, max_nb;
    double max_dist;
    double thr_norm;
    int has_prior;
    double beta_c, beta_e;
    double prev_b[3], prev_e[3];
    int normals;             // 0: the library default; NORMALS_EXACT / NORMALS_HYBRID / NORMALS_FAST (residual_jacobian)
};

struct alignas(8) GnState {
    double pose[14];         // begin (qx qy qz qw tx ty tz) | end
    double tbe[2];
    double slerp_theta, slerp_sin;
    int slerp_linear, slerp_negate;
    double x[12];
    double step_norm;
    int iter;                // solves completed
    int done;                // stop flag: converged, failed or error
    int failed;              // fewer than 100 keypoints contributed (ct_icp.cpp:860-871)
    int n_used;
    unsigned long long solve_cycles[4];   // shader clocks of the last k_reduce_solve: reduce | factorise | substitute | pose update
    // ICPSummary's avg_duration_* (include/ct_icp/ct_icp.h:164-168) without extra launches or events: the constant 100 MHz
    // wall clock (s_memrealtime) read by the first thread of the search kernel and at the start / end of the solve kernel
    unsigned long long clk_iter_start;    // written by the neighbour-search kernel of the running iteration
    unsigned long long ticks_neighborhood, ticks_solve, ticks_iter;   // sums over the executed iterations, 10 ns ticks
};

struct DebugView {
    int *n_nb;
    double *normal, *a2d, *farthest;
    uint8_t *used;
};

struct Counters {
    unsigned long long probed, hit, points;
};

// upper-triangular index tables of the packed system
__constant__ uint8_t c_tri_i[78] = {0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,1,1,1,1,1,1,1, 2,2,2,2,2,2,2,2,2,2, 3,3,3,3,3,3,3,3,3,
                                    4,4,4,4,4,4,4,4, 5,5,5,5,5,5,5, 6,6,6,6,6,6, 7,7,7,7,7, 8,8,8,8, 9,9,9, 10,10, 11};
__constant__ uint8_t c_tri_j[78] = {0,1,2,3,4,5,6,7,8,9,10,11, 1,2,3,4,5,6,7,8,9,10,11, 2,3,4,5,6,7,8,9,10,11,
                                    3,4,5,6,7,8,9,10,11, 4,5,6,7,8,9,10,11, 5,6,7,8,9,10,11, 6,7,8,9,10,11,
                                    7,8,9,10,11, 8,9,10,11, 9,10,11, 10,11, 11};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Quat st_qb(const GnState *s) { return {s->pose[0], s->pose[1], s->pose[2], s->pose[3]}; }
__device__ __forceinline__ Quat st_qe(const GnState *s) { return {s->pose[7], s->pose[8], s->pose[9], s->pose[10]}; }
__device__ __forceinline__ Vec3 st_tb(const GnState *s) { return {s->pose[4], s->pose[5], s->pose[6]}; }
__device__ __forceinline__ Vec3 st_te(const GnState *s) { return {s->pose[11], s->pose[12], s->pose[13]}; }

// pose_begin.InterpolatePose(pose_end, t) * raw (types.h:453-470, :360-366, :353-357)
__device__ __forceinline__ Vec3 ct_transform(const GnState *s, double alpha, Vec3 raw) {
    SlerpPair sp{s->slerp_theta, s->slerp_sin, s->slerp_linear, s->slerp_negate};
    Quat q = quat_normalized(slerp_eval(st_qb(s), st_qe(s), sp, alpha));
    Vec3 tb = st_tb(s), te = st_te(s);
    Vec3 r = quat_rotate(q, raw);
    double oma = 1.0 - alpha;
    return {r.x + (oma * tb.x + alpha * te.x), r.y + (oma * tb.y + alpha * te.y), r.z + (oma * tb.z + alpha * te.z)};
}

// hash lookup: returns block*128 + count (count >= 1) or 0 when the voxel is absent
__device__ __forceinline__ uint32_t map_lookup(const MapView &m, int vx, int vy, int vz) {
    uint64_t key = pack_key(vx, vy, vz);
    uint32_t i = hash_key(key, m.mask);
    for (;;) {
        Slot s = m.slots[i];
        if (s.key == key) return (s.block << 7) | s.count;
        if (s.key == KEY_EMPTY) return 0u;
        i = (i + 1) & m.mask;
    }
}

// A map point = 24 contiguous bytes (x y z, ctgn_map.hpp), 8-byte aligned: x and y come as ONE 16-byte load, z as an 8-byte one.
// `base + off` must address the point's x.
typedef double d2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void load_point(const char *base, uint32_t off, double &x, double &y, double &z) {
    const d2_a8 xy = *reinterpret_cast<const d2_a8 *>(base + off);
    x = xy.x; y = xy.y;
    z = *reinterpret_cast<const double *>(base + off + 16u);
}
constexpr uint32_t POINT_BYTES = 24u;

// Split lookup for software pipelining: probe_issue starts the first slot load of a voxel, probe_resolve finishes
// the probe sequence later (the load latency is covered by whatever runs in between).
struct Probe {
    uint64_t key;
    uint32_t idx;
    Slot s;
    bool active;
};
__device__ __forceinline__ Probe probe_issue(const MapView &m, bool active, int vx, int vy, int vz) {
    Probe p;
    p.active = active;
    p.key = pack_key(vx, vy, vz);
    p.idx = active ? hash_key(p.key, m.mask) : 0u;
    p.s = m.slots[p.idx];
    return p;
}
__device__ __forceinline__ uint32_t probe_resolve(const MapView &m, Probe &p) {
    if (!p.active) return 0u;
    for (;;) {
        if (p.s.key == p.key) return (p.s.block << 7) | p.s.count;
        if (p.s.key == KEY_EMPTY) return 0u;
        p.idx = (p.idx + 1) & m.mask;
        p.s = m.slots[p.idx];
    }
}

__device__ __forceinline__ bool sweep_in_short_range(int k, int nb) {
    // the reference's `short` sweep counters (map.h:470-472): outside this range its loops never end;
    // defined here (and in the oracle) as "no neighbours".
    return (k - nb >= -32768) && (k + nb + 1 <= 32767);
}

// How k_residual_reduce gets a neighbourhood's normal and a2D (neighborhood.h:285-316). In every mode the sums, the mean and the covariance
// are the reference build's (no fused multiply-add, its order); the modes differ in the 3 x 3 solver:
//   NORMALS_EXACT   Eigen's JacobiSVD restated operation for operation, correctly rounded division / square root (normal_a2d_exact,
//                   ctgn_math.hpp): normals and a2D bit-identical to the reference build's. Measured on the B2 sweep: residual kernel
//                   47 us instead of 30 (six IEEE divisions and three square roots per rotation), +11 % on the iteration.
//   NORMALS_HYBRID  (default) the fast cyclic Jacobi (sym3_normal_a2d: hardware reciprocal seeds + Newton steps). Both solvers are
//                   backward stable, so their normals differ by at most ~1e-15 |C| / (s1 - s2) <= 1e-15 / a2D^2; the exact one runs only
//                   for a keypoint whose gate `|dist| < max_dist` (ct_icp.cpp:803) lies within 1e-7 of its threshold under the fast normal,
//                   or whose a2D is below 1e-3 (normal decided by roundings: bound above > 1e-9). The discrete outputs — which keypoints
//                   pass the gate, n_used — are therefore the exact route's; normals and a2D agree with it to ~1e-12.
//   NORMALS_FAST    the fast solver only (rounds 1-3).
// GnParams::normals selects the mode (ctgn_set_normals, ct_icp_amd/csrc/ctgn_internal.h); bits 16-17 of the ablation mask override it (A/B).
constexpr int NORMALS_EXACT = 1, NORMALS_HYBRID = 2, NORMALS_FAST = 3;
// Wave-shared probes of the 125-voxel sweep (rows_tiles, shared2): 0 = off unless bit 22 of the ablation mask is set, 1 = on unless it is.
// OFF by default — and it always was: round 4 shipped the switch with its sense inverted (the macro was called ..._DEFAULT_OFF and compared
// with `!=`), so the "-2 %" of that round's A/B (D 0.857 -> 0.839 ms) was the gain of NOT sharing the probes, and rounds 4-5 measured every
// D number without them. Measured again in round 6 with the sense known (profiles/r06_ab_group_stage.txt, same box): D 0.6166 ms off,
// 0.6287 on (+2 %); C 0.0412 / 0.0411. Off stays.
#ifndef CTGN_SHARED2_DEFAULT_ON
#define CTGN_SHARED2_DEFAULT_ON 0
#endif
// Stream loop of the generic path: keep the next chunk's loads IN FRONT of the current chunk's test in the instruction stream. Left to
// itself the scheduler starts the test (whose first instructions wait for the current chunk) before it has issued the next chunk's
// loads, so only one chunk is ever in flight; with the fence the next chunk's loads are out before the wait.
// The probe-batch loop of the generic path (VIT = 2 batches of 16 sweep voxels on the 27-voxel sweep, 8 on the 125-voxel one).
#ifndef CTGN_BATCH_UNROLL
#define CTGN_BATCH_UNROLL _Pragma("unroll")
#endif
#ifndef CTGN_GUESS_PASSES
#define CTGN_GUESS_PASSES 2          // passes of a guessed first search: the guess, then the radius (3: twice the guessed distance in between — measured)
#endif
#ifndef CTGN_STREAM_DEPTH
#define CTGN_STREAM_DEPTH 2          // register sets of the stream loop. 3 (two chunks in flight) fits the registers without spills and
                                     // was measured: B2 0.1250 -> 0.1273 ms per iteration (the bounded searches stream 2-4 chunks per
                                     // batch: rounding those up to threes costs more than the deeper prefetch hides), D 0.7555 -> 0.7542
#endif
#ifndef CTGN_STREAM_FENCE
#define CTGN_STREAM_FENCE __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef CTGN_CULL1_DEFAULT
#define CTGN_CULL1_DEFAULT 0         // 1: the first search of a 27-voxel sweep culls its second probe batch's voxels (rows_tiles, cull1); bit 18 of the ablation mask flips it
#endif
#ifndef CTGN_NORMALS_DEFAULT
#define CTGN_NORMALS_DEFAULT 2
#endif
__device__ __forceinline__ int normals_mode(int ablate, int configured) { const int m = (ablate >> 16) & 3; return m ? m : (configured ? configured : CTGN_NORMALS_DEFAULT); }

// Gates + residual + 12-vector u for one keypoint (ct_icp.cpp:769-841). n = neighbours kept,
// S = sum p, SS = sum p p^T (6 unique), q = farthest kept neighbour (the reference's `closest_point`).
// The arithmetic between the sums and the gate is the reference build's: no fused multiply-add (a stock x86-64 build cannot fuse; the
// oracle is compiled -ffp-contract=off for the same reason), Eigen's association of the dot product at :782.
__device__ __forceinline__ bool residual_jacobian(int n, Vec3 S, Sym3 SS, Vec3 q, Vec3 p, Vec3 raw, double alpha,
                                                  const GnState *st, const GnParams &prm, int nmode, double u[12], double &r,
                                                  Vec3 &normal_out, double &a2d_out) {
#pragma clang fp contract(off)
    if (n < prm.min_nb || n < 5) return false;               // :769 ; neighborhood.h:227-230
    Vec3 nrm;
    double a2d;
    const Vec3 d = p - q;
    bool exact = nmode == NORMALS_EXACT;
    if (!exact) {
        const double dn = (double) n;
        const Vec3 mu{S.x / dn, S.y / dn, S.z / dn};         // neighborhood.h:241
        Sym3 C;                                              // :242-243
        C.xx = SS.xx / dn - mu.x * mu.x; C.xy = SS.xy / dn - mu.x * mu.y; C.xz = SS.xz / dn - mu.x * mu.z;
        C.yy = SS.yy / dn - mu.y * mu.y; C.yz = SS.yz / dn - mu.y * mu.z; C.zz = SS.zz / dn - mu.z * mu.z;
        sym3_normal_a2d(C, nrm, a2d);
        if (nmode == NORMALS_HYBRID) {
            const double dist_f = nrm.x * d.x + nrm.y * d.y + nrm.z * d.z;
            exact = !(a2d >= 1e-3) || fabs(fabs(dist_f) - prm.max_dist) <= 1e-7;
        }
    }
    if (exact) {
        const double SS9[9] = {SS.xx, SS.xy, SS.xz, SS.xy, SS.yy, SS.yz, SS.xz, SS.yz, SS.zz};
        normal_a2d_exact(n, S, SS9, nrm, a2d);               // neighborhood.h:236-244,293-311 in the reference build's arithmetic
    }
    const Vec3 tb = st_tb(st);
    if (nrm.x * (tb.x - p.x) + (nrm.y * (tb.y - p.y) + nrm.z * (tb.z - p.z)) < 0) nrm = Vec3{-nrm.x, -nrm.y, -nrm.z};      // :782-784
    normal_out = nrm;
    a2d_out = a2d;
    const double w = a2d * a2d;                                        // :787-788
    const Vec3 m{w * nrm.x, w * nrm.y, w * nrm.z};                     // :789
    const double dist = nrm.x * d.x + nrm.y * d.y + nrm.z * d.z;       // :793-795
    if (!(fabs(dist) < prm.max_dist)) return false;                    // :803
    r = m.x * d.x + m.y * d.y + m.z * d.z;                             // :805-807
    const Vec3 a = quat_rotate(st_qb(st), raw), e = quat_rotate(st_qe(st), raw);   // :813-816
    const double oma = 1.0 - alpha;
    u[0] = oma * (a.y * m.z - a.z * m.y); u[1] = oma * (a.z * m.x - a.x * m.z); u[2] = oma * (a.x * m.y - a.y * m.x);
    u[3] = oma * m.x; u[4] = oma * m.y; u[5] = oma * m.z;
    u[6] = alpha * (e.y * m.z - e.z * m.y); u[7] = alpha * (e.z * m.x - e.x * m.z); u[8] = alpha * (e.x * m.y - e.y * m.x);
    u[9] = alpha * m.x; u[10] = alpha * m.y; u[11] = alpha * m.z;
    return true;
}

// One entry of the packed system from a 13-double record (u[0..11], r) — entry e of 91.
__device__ __forceinline__ double sys_entry(const double *rec, int e) {
    if (e < 78) return rec[c_tri_i[e]] * rec[c_tri_j[e]];
    if (e < 90) return -rec[e - 78] * rec[12];
    return 0.0;
}

// ================================================================================================
// k_accumulate_lane — one lane per keypoint (cross-check kernel)
// ================================================================================================
constexpr int LANE_BLOCK = 128;

// ---- libstdc++'s std::priority_queue on (distance, payload) with the reference's comparator `a.distance < b.distance`
// (include/ct_icp/map.h:595-601), restated: push = push_back + std::push_heap, pop = std::pop_heap (+ pop_back) of bits/stl_heap.h.
// Which of two EQUAL distances a full queue keeps, and in which order it drains them, is decided by this layout and nothing else, so
// reproducing the reference on exact ties means replaying exactly these moves (the test oracle restates the same queue as its heap_mode 0).
// `ld(i)` / `st(i, item)` access slot i of the heap's storage (LDS here).
struct HeapItem {
    double d;          // the key: Euclidean distance (map.h:491)
    double s;          // its square (what the kernels carry around)
    uint32_t v;        // payload: visit index or point id
};
template <typename LD, typename ST>
__device__ __forceinline__ void heap_push_hole(LD ld, ST st, int hole, int top, const HeapItem &value) {
    int parent = (hole - 1) / 2;
    while (hole > top && ld(parent).d < value.d) {
        st(hole, ld(parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    st(hole, value);
}
template <typename LD, typename ST>
__device__ __forceinline__ void heap_push(LD ld, ST st, int &size, const HeapItem &value) {
    ++size;
    heap_push_hole(ld, st, size - 1, 0, value);
}
// std::pop_heap: the top moves to slot size-1, the rest is re-heaped; `size` shrinks by one. Draining a heap this way leaves the
// storage sorted ascending, equal keys in exactly the order the reference's drain loop (map.h:508-513) meets them, reversed.
template <typename LD, typename ST>
__device__ __forceinline__ void heap_pop(LD ld, ST st, int &size) {
    if (size > 1) {
        const HeapItem value = ld(size - 1);
        st(size - 1, ld(0));
        const int len = size - 1;
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (ld(child).d < ld(child - 1).d) child--;
            st(hole, ld(child));
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            st(hole, ld(child - 1));
            hole = child - 1;
        }
        heap_push_hole(ld, st, hole, 0, value);
    }
    --size;
}

// One lane, one query: the reference's RadiusSearchInPlace verbatim (map.h:449-514) — x-major voxel sweep, insertion order inside a
// voxel, `distance > radius` skip, bounded std::priority_queue with strict-< replacement, drain. The queue lives in LDS (column `tid`
// of d2s / ids; d2s holds DISTANCES, as the reference compares them). On return the n kept entries are sorted ascending in place
// ([0] nearest ... [n-1] farthest), ties in the reference's drain order: its neighbour j (farthest first) is entry n-1-j.
__device__ __forceinline__ int lane_search(const MapView &m, Vec3 p, int k, double *d2s, uint32_t *ids, int stride,
                                           Counters *cnt_out) {
    int kx = voxel_coord(p.x, m.resolution), ky = voxel_coord(p.y, m.resolution), kz = voxel_coord(p.z, m.resolution);
    if (!(sweep_in_short_range(kx, m.nb) && sweep_in_short_range(ky, m.nb) && sweep_in_short_range(kz, m.nb))) return 0;
    int n = 0;
    unsigned long long c_probe = 0, c_hit = 0, c_pts = 0;
    auto ld = [&](int i) { return HeapItem{d2s[i * stride], 0.0, ids[i * stride]}; };
    auto st = [&](int i, const HeapItem &it) { d2s[i * stride] = it.d; ids[i * stride] = it.v; };
    for (int vx = kx - m.nb; vx <= kx + m.nb; ++vx)
        for (int vy = ky - m.nb; vy <= ky + m.nb; ++vy)
            for (int vz = kz - m.nb; vz <= kz + m.nb; ++vz) {
                uint32_t bc = map_lookup(m, vx, vy, vz);
                c_probe++;
                if (!bc) continue;
                uint32_t block = bc >> 7, count = bc & 127u;
                c_hit++;
                c_pts += count;
                const double *bx = m.blocks + (size_t) block * 3 * m.blk;
                for (uint32_t i = 0; i < count; ++i) {
                    double dx = bx[3 * i] - p.x, dy = bx[3 * i + 1] - p.y, dz = bx[3 * i + 2] - p.z;
                    double d2 = sq_norm3(dx, dy, dz);
                    if (d2 > m.r2thr) continue;                       // map.h:491-493: sqrt(d2) > radius
                    const HeapItem it{__dsqrt_rn(d2), d2, block * (uint32_t) m.blk + i};
                    if (n == k) {                                     // map.h:494-500
                        if (it.d < d2s[0]) { heap_pop(ld, st, n); heap_push(ld, st, n, it); }
                    } else {
                        heap_push(ld, st, n, it);
                    }
                }
            }
    const int kept = n;
    for (int sz = kept; sz > 1;) heap_pop(ld, st, sz);                // map.h:508-513
    if (cnt_out) {
        atomicAdd(&cnt_out->probed, c_probe);
        atomicAdd(&cnt_out->hit, c_hit);
        atomicAdd(&cnt_out->points, c_pts);
    }
    return kept;
}

__device__ __forceinline__ Vec3 map_point(const MapView &m, uint32_t id) {
    uint32_t block = id / (uint32_t) m.blk, i = id - block * (uint32_t) m.blk;
    const double *bx = m.blocks + (size_t) block * 3 * m.blk;
    return {bx[3 * i], bx[3 * i + 1], bx[3 * i + 2]};
}

__global__ __launch_bounds__(LANE_BLOCK) void k_accumulate_lane(MapView map, KpView kp, const GnState *st, GnParams prm,
                                                                double *partials, DebugView dbg, int first_iter) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && kp.clk_iter_start) *kp.clk_iter_start = wall_clock64();
    const int tid = threadIdx.x;
    const int k = prm.max_nb;
    double *d2s = reinterpret_cast<double *>(smem);                        // [k][LANE_BLOCK]
    uint32_t *ids = reinterpret_cast<uint32_t *>(d2s + (size_t) KMAX * LANE_BLOCK);   // [k][LANE_BLOCK]
    double *rec = reinterpret_cast<double *>(ids + (size_t) KMAX * LANE_BLOCK);       // [LANE_BLOCK][13]
    int *wcnt = reinterpret_cast<int *>(rec + (size_t) LANE_BLOCK * 13);               // [LANE_BLOCK/64]
    double acc = 0.0;
    const int ntiles = (kp.n + LANE_BLOCK - 1) / LANE_BLOCK;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int i = tile * LANE_BLOCK + tid;
        double u[12], r = 0.0;
        bool used = false;
        if (i < kp.n) {
            Vec3 raw{kp.rx[i], kp.ry[i], kp.rz[i]};
            double alpha = alpha_timestamp(kp.t[i], st->tbe[0], st->tbe[1]);
            Vec3 p;
            if (first_iter) {
                p = Vec3{kp.wx[i], kp.wy[i], kp.wz[i]};                     // ct_icp.cpp:756 (iteration 0 uses the input)
            } else {
                p = ct_transform(st, alpha, raw);                          // :964-966 of the previous iteration
                kp.wx[i] = p.x; kp.wy[i] = p.y; kp.wz[i] = p.z;
            }
            int n = lane_search(map, p, k, d2s + tid, ids + tid, LANE_BLOCK, nullptr);
            Vec3 S{0, 0, 0}, q{0, 0, 0};
            Sym3 SS{0, 0, 0, 0, 0, 0};
            for (int j = n - 1; j >= 0; --j) {                              // farthest first: the reference's neighbour order (map.h:508-513),
#pragma clang fp contract(off)
                Vec3 c = map_point(map, ids[j * LANE_BLOCK + tid]);         // which is the order ComputeNeighborhood sums in (neighborhood.h:236-240)
                S.x += c.x; S.y += c.y; S.z += c.z;
                SS.xx += c.x * c.x; SS.xy += c.x * c.y; SS.xz += c.x * c.z;
                SS.yy += c.y * c.y; SS.yz += c.y * c.z; SS.zz += c.z * c.z;
                if (j == n - 1) q = c;                                      // points[0]: the farthest kept (ct_icp.cpp:791)
            }
            Vec3 nrm{0, 0, 0};
            double a2d = 0.0;
            used = residual_jacobian(n, S, SS, q, p, raw, alpha, st, prm, normals_mode(0, prm.normals), u, r, nrm, a2d);
            if (dbg.n_nb) {
                dbg.n_nb[i] = n;
                dbg.normal[3 * i] = nrm.x; dbg.normal[3 * i + 1] = nrm.y; dbg.normal[3 * i + 2] = nrm.z;
                dbg.a2d[i] = a2d;
                dbg.farthest[3 * i] = q.x; dbg.farthest[3 * i + 1] = q.y; dbg.farthest[3 * i + 2] = q.z;
                dbg.used[i] = used ? 1 : 0;
            }
        }
        __syncthreads();                      // everyone is done with the search lists of this tile
        double *my = rec + tid * 13;
        for (int c = 0; c < 12; ++c) my[c] = used ? u[c] : 0.0;
        my[12] = used ? r : 0.0;
        const unsigned long long ub = ballot64(used);
        if ((tid & 63) == 0) wcnt[tid >> 6] = __popcll(ub);
        __syncthreads();
        if (tid < 90) {
            for (int j = 0; j < LANE_BLOCK; ++j) acc += sys_entry(rec + j * 13, tid);
        } else if (tid == 90) {
            for (int w = 0; w < LANE_BLOCK / 64; ++w) acc += (double) wcnt[w];
        }
        __syncthreads();
    }
    if (tid < SYS_N) partials[(size_t) blockIdx.x * SYS_N + tid] = (tid < SYS_USED) ? acc : 0.0;
}

inline size_t lane_kernel_smem() {
    return (size_t) KMAX * LANE_BLOCK * (sizeof(double) + sizeof(uint32_t)) + (size_t) LANE_BLOCK * 13 * sizeof(double) + 64;
}

// ================================================================================================
// k_accumulate_rows — 16 lanes per keypoint
// ================================================================================================
constexpr int ROW_WAVES = 4;                 // waves per block
constexpr int ROW_BLOCK = ROW_WAVES * 64;
#ifndef CTGN_LCAP
#define CTGN_LCAP 96
#endif
constexpr int LCAP = CTGN_LCAP;               // per-row candidate list capacity (>= KMAX + 2*16)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// all-reduce over the 16 lanes of a DPP row: xor-1, xor-2 (quad_perm), then half-mirror and mirror, which act
// as xor-4 / xor-8 once the lanes of a quad / half-row already agree. Fixed order -> deterministic.
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);     // row_half_mirror
    v += dpp_f64<0x140>(v);     // row_mirror
    return v;
}
__device__ __forceinline__ int row_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
    return v;
}
__device__ __forceinline__ int row_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
    return v;
}
// inclusive prefix sum over the 16 lanes of a row (row_shr with zero fill)
__device__ __forceinline__ int row_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    return v;
}

// Order in which the row kernel probes the (2 NB + 1)^3 sweep voxels: by squared offset from the query's voxel
// (centre, faces, edges, corners, ...), ties by the reference's x-major sweep index. The VALUES are those sweep
// indices v = (ox+NB)*S*S + (oy+NB)*S + (oz+NB): they — not the probing order — define the visit order used for
// tie-breaking, so any probing order yields the reference's result. 255 pads the last batch of 16.
template <int NB>
struct SweepOrder {
    uint8_t v[(((2 * NB + 1) * (2 * NB + 1) * (2 * NB + 1) + 15) / 16) * 16];
};
template <int NB>
constexpr SweepOrder<NB> make_sweep_order() {
    constexpr int S = 2 * NB + 1, V = S * S * S, PAD = ((V + 15) / 16) * 16;
    SweepOrder<NB> t{};
    int key[PAD] = {};
    for (int i = 0; i < PAD; ++i) { t.v[i] = 255; key[i] = 1 << 30; }
    for (int v = 0; v < V; ++v) {
        const int ox = v / (S * S) - NB, oy = (v / S) % S - NB, oz = v % S - NB;
        const int kv = (ox * ox + oy * oy + oz * oz) * 256 + v;
        int pos = v;                                   // insertion sort
        while (pos > 0 && key[pos - 1] > kv) { key[pos] = key[pos - 1]; t.v[pos] = t.v[pos - 1]; --pos; }
        key[pos] = kv;
        t.v[pos] = (uint8_t) v;
    }
    return t;
}
__constant__ SweepOrder<1> c_sweep1 = make_sweep_order<1>();
__constant__ SweepOrder<2> c_sweep2 = make_sweep_order<2>();

// Distance from q to the interval of coordinates that Voxel::Coordinates maps to index `vox` (int(p / res) truncates
// toward zero, so index 0 is (-res, res) and a negative index v is ((v-1) res, v res]; src/SlamCore/types.cxx:13-20).
__device__ __forceinline__ double axis_gap(double q, int vox, double res) {
    // index v > 0: [v res, (v + 1) res); v < 0: ((v - 1) res, v res]; 0: (-res, res). Branch-free (the three-way branch this replaces was
    // divergent in every wave that straddles an axis plane and cost two exec save / restore pairs per axis and probe); the callers' 1e-8
    // relative slack covers the last-bit difference between (v + 1) res and v res + res.
    const int lo_i = vox - (vox <= 0 ? 1 : 0), hi_i = vox + (vox >= 0 ? 1 : 0);
    const double lo = (double) lo_i * res, hi = (double) hi_i * res;
    return fmax(fmax(lo - q, q - hi), 0.0);
}

// Issue the hash probe of this lane's voxel of probe batch `it`: sweep index v (255 = none), skipped when the
// keypoint has no search or when the whole voxel lies farther from the query than r2bound — the radius, or the tighter
// bound on the k-th neighbour's distance carried over from the previous search (exact: no point of such a voxel can be
// among the k nearest within the radius, map.h:491-493; a 1e-8 relative slack covers the rounding of the voxel boundaries).
template <int NB>
__device__ __forceinline__ bool batch_reach(const MapView &m, int it, int sub, bool searching, int kx, int ky, int kz,
                                            double qx, double qy, double qz, int &v_out, int &vx, int &vy, int &vz, int ablate, double r2bound,
                                            uint32_t mreach) {
    constexpr int S = 2 * NB + 1;
    const int v = (NB == 1) ? (int) c_sweep1.v[it * 16 + sub] : (int) c_sweep2.v[it * 16 + sub];
    v_out = v;
    const int vv = (v == 255) ? 0 : v;
    const int ox = vv / (S * S) - NB, oy = (vv / S) % S - NB, oz = vv % S - NB;
    vx = kx + ox; vy = ky + oy; vz = kz + oz;
    // per-axis slab test first (the keypoint's reach mask, phase A: three bit tests): a voxel outside the box of reachable slabs is
    // outside the sphere; only the others pay for the exact test. With a carried-over bound most of a 125-voxel sweep stops here.
    bool reachable = searching && v != 255 &&
                     ((mreach >> (ox + 2)) & (mreach >> (5 + oy + 2)) & (mreach >> (10 + oz + 2)) & 1u) != 0u;
    if (reachable && !(ablate & 64)) {
        const double gx = axis_gap(qx, vx, m.resolution), gy = axis_gap(qy, vy, m.resolution), gz = axis_gap(qz, vz, m.resolution);
        reachable = gx * gx + gy * gy + gz * gz <= r2bound * (1.0 + 1e-8) + 1e-12;
    }
    return reachable;
}
template <int NB>
__device__ __forceinline__ Probe issue_batch(const MapView &m, int it, int sub, bool searching, int kx, int ky, int kz,
                                             double qx, double qy, double qz, int &v_out, int ablate, double r2bound, uint32_t mreach) {
    int vx, vy, vz;
    const bool reachable = batch_reach<NB>(m, it, sub, searching, kx, ky, kz, qx, qy, qz, v_out, vx, vy, vz, ablate, r2bound, mreach);
    return probe_issue(m, reachable && !(ablate & 16), vx, vy, vz);
}

// the 16 ballot bits of DPP row `row`
__device__ __forceinline__ uint32_t row_bits(unsigned long long ballot, int row) {
    return (uint32_t) (ballot >> (16 * row)) & 0xffffu;
}
// max over the 4 rows of a row-uniform value, as a wave-uniform scalar (readlane -> SALU, no LDS round trip)
__device__ __forceinline__ int max_over_rows(int v) {
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// per-row candidate list (always live during phase B)
template <int CAP>
struct RowListT {
    static constexpr int cap = CAP;
    double d2[CAP];
    uint32_t vis[CAP];        // (sweep voxel index << 6) | slot : the visit order, and the way back to the point
    uint32_t hist[16];
};
using RowList = RowListT<LCAP>;
constexpr int PCAP = 48;                     // list capacity of the pool-check kernel: KMAX entries + the one a selection parks behind them, in whole rows of 16
using PoolList = RowListT<PCAP>;
// per-row probe scratch of the generic path
template <int OCC>
struct RowProbe {
    uint32_t occ[OCC];        // block*128 + count per sweep voxel of this row's keypoint
    uint2 chunk[64];          // per 16-point chunk of the probe batch in flight: .x = byte offset of its first x,
                              // .y = (visit index of its first point << 8) | points in the chunk (1..16)
};
// wave-shared neighbourhood of the fast path (NB = 1, BLK <= 32): when the four keypoints of a round live in the same
// home voxel their 27-voxel neighbourhood is probed ONCE per wave and flattened into a dense candidate table, so the
// rows stream 16 real candidates per step (no per-voxel padding) and all rows run the same number of steps. The table
// stays valid for the following rounds with the same home voxel (keypoints are sorted by home voxel on upload).
constexpr int STAGE_CAP = 27 * 32;
struct SharedStage {
    uint16_t vis[STAGE_CAP];  // its visit index (sweep voxel << 6) | slot
    uint32_t occ[28];         // block*128 + count per sweep voxel
    uint2 chunk[56];          // .x = byte offset of the chunk's first x; .y = (visit base << 15) | (flat position << 5) | points
};

template <int OCC>
struct WaveScratch {
    double px[64], py[64], pz[64];     // world point of the tile's keypoints
    int kx[64], ky[64], kz[64];        // its voxel; kx == INT_MIN -> no search
    int id[64];                        // its index in the caller's arrays; -1 = none
    float kb[64];                      // admission bound of its search (squared distance, rounded up); +inf = the radius only
    float rr2[64];                     // pool certificate: every map point NOT in its pool is at least sqrt(rr2) away from the new position (0: no pool)
    uint16_t mr[64];                   // per axis, which voxel offsets -2 .. +2 of its home voxel reach inside that bound (bit 5 a + o + 2)
    uint8_t todo[64];                  // 1: the keypoint needs a search this iteration (no pool, or its pool could not be certified)
    uint8_t slot[64];                  // round r, row j of the search phase works on the keypoint of lane slot[4 r + j] (255: nothing)
    uint8_t gs[64];                    // 1: its bound (kb) is a GUESS, not a proven upper bound of its k-th neighbour's distance (KpView::guess2)
    uint32_t socc[OCC >= 128 ? 128 : 4];   // 125-voxel sweep: block*128 + count of every sweep voxel of the round's SHARED home voxel, probed once per wave
    union {
        struct {
            RowList list[4];
            union {
                RowProbe<OCC> probe[4];
                SharedStage stage;
            };
        };
        double rec[64 * 13];           // phase D: u[12] | r per keypoint
    };
};

// Home-voxel-group stage of the 125-voxel sweep (round 6, prototype behind the tuning key `stage_lds`; template flag STAGE of rows_tiles).
// On an upload sorted by home voxel (config D: ~100 keypoints per 0.5 m voxel) consecutive rounds of a chunked tile (KpView::chunk) search
// the same handful of voxels. The wave copies the points of the home voxel's INNER neighbourhood (the <= 27 voxels within +-1 per axis
// that some keypoint of the group can reach) into LDS once — flat, nearest voxels first, each point with its visit index — and every row
// of every round of the group then streams that table: no reach tests, no chunk lists, no global loads and no address arithmetic in the
// stream loop, and all four rows run the same number of steps. A point of a voxel a row's bound does not reach fails `d2 <= bound` like
// any other (the reach test is a superset test of exactly that), so the admitted set — and every result — is the generic path's.
// 7.3 KB per wave on top of WaveScratch's 12.9: two blocks per CU instead of three (the instantiation is compiled for 2 waves per SIMD).
constexpr int GCAP = 272;                    // points the table holds; a group that needs more takes the generic path
struct GroupStage {
    double x[GCAP], y[GCAP], z[GCAP];
    uint16_t vis[GCAP];                      // (sweep voxel index << 6) | slot, as RowList::vis
    uint2 vinfo[28];                         // inner voxel i (nearest first): .x = byte offset of its block, .y = (sweep index << 20) | (flat start << 8) | points
};

// k nearest of the row's list (d2, vis)[0..Ln) under the total order (d2, vis); winners are written back
// sorted ascending at [0..min(Ln,k)). HIST: first cut the list with a 16-bin histogram of d2 over [0, hi]
// (one bin per lane): only the bins up to the one in which the running count reaches k can hold winners.
// All loops run to the wave-uniform maximum over the 4 rows; loads are unconditional (indices stay inside the
// arrays) and masked afterwards, so the code is branch-light.
// near_tie (out, row-uniform): some entry that is kept — or is the first one dropped — lies within a few ulps of another entry's
// squared distance. The total order (d2, visit index) and the reference's std::priority_queue keyed by sqrt(d2) (map.h:491-500) can
// then disagree on which entry is kept and on the order of the kept ones; the caller replays the reference's queue for that keypoint
// (replay_reference_queue). Only the exact rank below can see such a pair: two float32 keys that differ are no near-tie.
constexpr double NEAR_TIE_REL = 0x1p-49;       // sqrt maps at most three adjacent doubles to one: |a - b| <= 3 ulp; 2^-49 is 8-16 ulp
// bound that admits everything the reference could treat as tying the current k-th best: candidates up to a few ulps ABOVE its
// squared distance may have the same sqrt, and one the reference visited EARLIER than the current k-th would have been kept by it
__device__ __forceinline__ double kth_bound(double kth_sq, double r2thr) { return fmin(r2thr, kth_sq * (1.0 + 0x1p-48)); }

template <bool HIST, typename RL>
__device__ __forceinline__ int row_select(RL &R, int Ln, int k, int sub, int row, double hi, bool &near_tie) {
    constexpr int MAXOWN = RL::cap / 16;         // entries a lane can own (the caller keeps Ln <= RL::cap)
    int maxLn = max_over_rows(Ln);
    if (maxLn == 0) return 0;
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const uint32_t lt_mask = (1u << sub) - 1u;
    // lists of at most 32 entries go straight to the register rank below (with a carried-over bound that is nearly every list);
    // longer ones are first cut with the histogram
    // Up to three passes: the first bins [0, hi]; a pass that still leaves a row more than 32 entries (many candidates in the bin
    // the k-th one falls into) bins that one bin again, entries below it kept as they are.
    double lo = 0.0, width = hi;
    auto hist_pass = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;      // lo == 0: no entries below the binned range
        R.hist[sub] = 0;
        const double scale = width > 0.0 ? 16.0 / width : 0.0;
        const int nown = (maxLn + 15) >> 4;
        const bool cut = Ln > k;
        auto bin_of = [&](double d) { return FIRST ? min(15, (int) (d * scale)) : (d < lo ? -1 : min(15, (int) ((d - lo) * scale))); };
        int below = 0;                          // entries under the binned range: kept, and counted towards k
        for (int m = 0; m < nown; ++m) {
            const int e = sub + 16 * m;
            const int b = bin_of(R.d2[e]);
            if (cut && e < Ln && b >= 0) atomicAdd(&R.hist[b], 1u);
            if (!FIRST) below += __popc(row_bits(ballot64(cut && e < Ln && b < 0), row));
        }
        const int cum = below + row_scan_i32((int) R.hist[sub]);
        const uint32_t reach = row_bits(ballot64(cum >= k), row);
        const int bb = reach ? (__ffs(reach) - 1) : 15;
        // stable in-place compaction of the entries with bin <= bb: step m reads 16 entries, then writes at
        // positions <= the ones it read (LDS operations of a wave execute in order)
        int base = 0;
        for (int m = 0; m < nown; ++m) {
            const int e = sub + 16 * m;
            const double d = R.d2[e];
            const uint32_t vv = R.vis[e];
            // (entries a hair above the last kept bin stay too: a near-tie of the k-th best must not be cut away unseen)
            const bool keep = (e < Ln) && (!cut || bin_of(d * (1.0 - 0x1p-40)) <= bb);
            const uint32_t km = row_bits(ballot64(keep), row);
            if (keep) {
                const int pos = base + __popc(km & lt_mask);
                R.d2[pos] = d;
                R.vis[pos] = vv;
            }
            base += __popc(km);
        }
        // a further pass bins the last kept bin again (each row its own)
        if (cut) { lo = lo + bb * (width / 16.0); width = width / 16.0; }
        Ln = base;
        maxLn = max_over_rows(Ln);
    };
    if constexpr (HIST && RL::cap > 48) {        // (a pool-check list never exceeds 32 entries)
        if (maxLn > 32) {
            hist_pass(std::true_type{});
            for (int pass = 1; pass < 3 && maxLn > 32; ++pass) hist_pass(std::false_type{});
        }
    }
    if (HIST && maxLn <= 32) {
        // Fast rank for <= 32 surviving entries per row: every lane owns entries `sub` and `sub + 16` and keeps their
        // float32 keys (monotone roundings of d2) in registers; the keys are rotated round the DPP row 15 times and
        // counted against the owned ones — no LDS traffic. float32 cannot separate every pair of doubles: a key that
        // ties another one sends the whole wave to the exact rank below (rare).
        const float FINF = __int_as_float(0x7f800000);
        const int e0 = sub, e1 = sub + 16;
        const double d0 = R.d2[e0], d1 = R.d2[e1];
        const uint32_t v0 = R.vis[e0], v1 = R.vis[e1];
        const float k0 = e0 < Ln ? (float) d0 : FINF, k1 = e1 < Ln ? (float) d1 : FINF;
        int lt0 = (k1 < k0) ? 1 : 0, lt1 = (k0 < k1) ? 1 : 0;       // the lane's own pair
        int scratch_;
        // The other 15 lanes' keys are read straight from their registers: the DPP selector row_ror:S on the first source delivers
        // lane (sub + S) mod 16 of the row — no rotating copies, no moves. The keys are non-negative floats (squared distances, +inf
        // padding), which order like their bit patterns as unsigned integers, so "other < mine" is the borrow of an unsigned subtract
        // (VOP2, which takes a DPP source; compares do not on this part) and goes into the rank with one add-with-carry. Hand-placed
        // (the compiler materialises every rotated key with a v_mov_dpp and pairs compares through v_cndmask): 8 instructions per step
        // instead of 12. s_nop 1: VALU write of VCC -> VALU read of VCC as carry-in needs two wait states on this part (the compiler
        // puts the same s_nop 1 between its own v_sub_co / v_subb pairs; inline assembly is not seen by its hazard recogniser).
#define CTGN_RANK_CMP(S, OTHER, MINE, RANK)                                                                                       \
        asm volatile("v_sub_co_u32_dpp %1, vcc, %2, %3 row_ror:" #S " row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" \
                     : "+v"(RANK), "=&v"(scratch_) : "v"(OTHER), "v"(MINE) : "vcc");
#define CTGN_RANK_STEP(S) CTGN_RANK_CMP(S, u0, u0, lt0) CTGN_RANK_CMP(S, u1, u0, lt0) CTGN_RANK_CMP(S, u0, u1, lt1) CTGN_RANK_CMP(S, u1, u1, lt1)
        const int u0 = __float_as_int(k0), u1 = __float_as_int(k1);
        asm volatile("s_nop 4" ::: "memory");       // k0 / k1 were just written by the VALU (and EXEC may have been): DPP reads need the wait states
        CTGN_RANK_STEP(1) CTGN_RANK_STEP(2) CTGN_RANK_STEP(3) CTGN_RANK_STEP(4) CTGN_RANK_STEP(5) CTGN_RANK_STEP(6) CTGN_RANK_STEP(7)
        CTGN_RANK_STEP(8) CTGN_RANK_STEP(9) CTGN_RANK_STEP(10) CTGN_RANK_STEP(11) CTGN_RANK_STEP(12) CTGN_RANK_STEP(13) CTGN_RANK_STEP(14)
        CTGN_RANK_STEP(15)
#undef CTGN_RANK_STEP
#undef CTGN_RANK_CMP
        // Ties without a second set of counters: over the Ln real keys the strict ranks add up to Ln (Ln - 1) / 2 iff no two keys are
        // equal (every tying pair is missing from the sum). Padding keys (+inf) are never smaller than anything, so the real keys' ranks
        // only count real keys.
        const int rsum = row_sum_i32((e0 < Ln ? lt0 : 0) + (e1 < Ln ? lt1 : 0));
        const bool tie = rsum != (Ln * (Ln - 1)) / 2;
        if (!any64(tie)) {
            if (e0 < Ln && lt0 < k) { R.d2[lt0] = d0; R.vis[lt0] = v0; }
            if (e1 < Ln && lt1 < k) { R.d2[lt1] = d1; R.vis[lt1] = v1; }
            return Ln < k ? Ln : k;
        }
    }
    // rank sort: rank(e) = #{f : key_f < key_e}; ranks are distinct, winners land at their rank
    double od2[MAXOWN];
    uint32_t ovis[MAXOWN];
    int rank[MAXOWN];
#pragma unroll
    for (int m = 0; m < MAXOWN; ++m) {
        const int e = sub + 16 * m;
        const bool ok = e < Ln;
        const double d = R.d2[e];
        const uint32_t vv = R.vis[e];
        od2[m] = ok ? d : INF;
        ovis[m] = ok ? vv : 0xffffffffu;
        rank[m] = 0;
    }
    const int mcount = (maxLn + 15) >> 4;
#pragma unroll 4
    for (int f = 0; f < maxLn; ++f) {
        const bool fv = f < Ln;
        const double fdr = R.d2[f];
        const uint32_t fvr = R.vis[f];
        const double fd2 = fv ? fdr : INF;
        const uint32_t fvis = fv ? fvr : 0xffffffffu;
#pragma unroll
        for (int m = 0; m < MAXOWN; ++m) {
            if (m < mcount) rank[m] += (fd2 < od2[m] || (fd2 == od2[m] && fvis < ovis[m])) ? 1 : 0;
        }
    }
    // every row is written back sorted (rows with <= k entries keep them all); the first entry that is DROPPED (rank k) goes right
    // behind the kept ones (slot k: outside the list the caller sees) so that the tie check below can look at it
#pragma unroll
    for (int m = 0; m < MAXOWN; ++m) {
        const int e = sub + 16 * m;
        if (e < Ln && rank[m] <= k) { R.d2[rank[m]] = od2[m]; R.vis[rank[m]] = ovis[m]; }
    }
    {
        // (near-)ties: adjacent entries of the sorted run [0 .. min(Ln, k + 1)) within NEAR_TIE_REL of each other — a pair of kept
        // entries, or the k-th kept and the first dropped one. Two pairs per lane cover k <= 32.
        const int lim = min(Ln, k + 1);
        const double a0 = R.d2[sub], b0 = R.d2[sub + 1], a1 = R.d2[sub + 16], b1 = R.d2[sub + 17];
        const bool mine = (sub + 1 < lim && b0 - a0 <= b0 * NEAR_TIE_REL) || (sub + 17 < lim && b1 - a1 <= b1 * NEAR_TIE_REL);
        near_tie = near_tie || row_bits(ballot64(mine), row) != 0u;
    }
    return Ln < k ? Ln : k;
}

// ---- keypoints whose candidates (nearly) tie: the reference's queue decides, so it is replayed — by the kernels that CONSUME the
// neighbour records (k_residual_reduce, k_robust_prepare), not by the search kernel, whose registers and instruction stream stay as
// they are (an inlined replay cost the search kernel 4 % on the B2 sweep). The search kernel only sets TIE_FLAG in the record's count.
// The distances the searches carry from one iteration to the next (KpView::kth) are BOUNDS: a k-th neighbour's distance rounded up, a pool's
// completeness radius rounded down. They used to be taken with a double-precision square root (a ~30-instruction sequence, twice per hand-over
// and per pool check: profiles/r06_search_kernel_isa_budget.txt); one v_sqrt_f32 (1 ulp) of the directed float and two float ulps of margin
// bound the same quantity. Any valid bound leaves the k nearest — and every result — unchanged.
__device__ __forceinline__ float sqrt_bound_up(double x) {
    return __int_as_float(__float_as_int(__builtin_amdgcn_sqrtf(__double2float_ru(x))) + 2);
}
__device__ __forceinline__ float sqrt_bound_down(double x) {
    return __int_as_float(max(__float_as_int(__builtin_amdgcn_sqrtf(__double2float_rd(x))) - 2, 0));
}
constexpr uint32_t TIE_FLAG = 0x80000000u;
constexpr uint32_t REC_N_MASK = 63u;           // record word 0: bits 0-5 neighbours kept (n), bits 8-13 pool size (m >= n), bit 31 TIE_FLAG
#ifndef CTGN_POOL_REFILL
#define CTGN_POOL_REFILL 4
#endif
constexpr int POOL_REFILL = CTGN_POOL_REFILL;                 // a keypoint whose pool check fails is searched up to its (k + POOL_REFILL)-th pool member
constexpr int POOL_EXTRA = 8;                  // spare pool members behind the k neighbours (k + POOL_EXTRA <= KMAX or as many as fit)
struct TieScratch {            // one per wave: the queue of the lane being replayed (keys, their squares, payload = point byte offset)
    double d[KMAX], s[KMAX];
    uint32_t v[KMAX];
};
// The reference's RadiusSearchInPlace replayed literally for ONE keypoint by ONE lane (map.h:449-514: x-major sweep of all
// (2 nb + 1)^3 voxels, insertion order inside a voxel, `distance > radius` skip, bounded std::priority_queue keyed by the distance,
// drain). On return T.v[0 .. n) are the byte offsets of the kept points nearest first, ties in the reference's drain order (its
// neighbour j, farthest first, is entry n-1-j). Slow (27 / 125 dependent probes by one lane) and rare: real scans never tie; a lattice
// map does at every query.
__device__ __forceinline__ int replay_reference_queue(const MapView &m, Vec3 q, int k, TieScratch &T) {
    const int kx = voxel_coord(q.x, m.resolution), ky = voxel_coord(q.y, m.resolution), kz = voxel_coord(q.z, m.resolution);
    if (!(sweep_in_short_range(kx, m.nb) && sweep_in_short_range(ky, m.nb) && sweep_in_short_range(kz, m.nb))) return 0;
    auto ld = [&](int i) { return HeapItem{T.d[i], T.s[i], T.v[i]}; };
    auto st = [&](int i, const HeapItem &it) { T.d[i] = it.d; T.s[i] = it.s; T.v[i] = it.v; };
    const uint32_t stride3 = 3u * (uint32_t) m.blk * 8u;
    int n = 0;
    for (int vx = kx - m.nb; vx <= kx + m.nb; ++vx)
        for (int vy = ky - m.nb; vy <= ky + m.nb; ++vy)
            for (int vz = kz - m.nb; vz <= kz + m.nb; ++vz) {
                const uint32_t bc = map_lookup(m, vx, vy, vz);
                if (!bc) continue;
                const uint32_t count = bc & 127u, base = (bc >> 7) * stride3;
                const double *bx = m.blocks + (size_t) (bc >> 7) * 3 * m.blk;
                for (uint32_t i = 0; i < count; ++i) {
                    const double dx = bx[3 * i] - q.x, dy = bx[3 * i + 1] - q.y, dz = bx[3 * i + 2] - q.z;
                    const double d2 = sq_norm3(dx, dy, dz);
                    if (d2 > m.r2thr) continue;                        // map.h:491-493
                    const HeapItem it{__dsqrt_rn(d2), d2, base + i * POINT_BYTES};
                    if (n == k) {                                      // map.h:494-500
                        if (it.d < T.d[0]) { heap_pop(ld, st, n); heap_push(ld, st, n, it); }
                    } else {
                        heap_push(ld, st, n, it);
                    }
                }
            }
    const int kept = n;
    for (int sz = kept; sz > 1;) heap_pop(ld, st, sz);                 // map.h:508-513: [kept-1] is drained first ... [0] last
    return kept;
}
// All (active) lanes of a wave call this right after loading their keypoint's record into rec32 (count | TIE_FLAG, then the byte
// offsets nearest first). Flagged lanes are replayed one after the other through the wave's TieScratch and their record replaced.
__device__ __forceinline__ void resolve_ties(const MapView &map, const KpView &kp, int my_kp, bool wanted, uint32_t (&rec32)[SEL_STRIDE],
                                             TieScratch &T, int lane, int k) {
    unsigned long long todo = ballot64(wanted && (rec32[0] & TIE_FLAG) != 0u);
    while (todo) {
        const int L = __ffsll((long long) todo) - 1;
        todo &= todo - 1ull;
        if (lane == L) {
            const int n = replay_reference_queue(map, Vec3{kp.wx[my_kp], kp.wy[my_kp], kp.wz[my_kp]}, k, T);
            rec32[0] = (uint32_t) n;
#pragma unroll
            for (int q = 0; q < KMAX; ++q) rec32[1 + q] = q < n ? T.v[q] : 0u;
        }
    }
    rec32[0] &= REC_N_MASK;
}

// true iff the four rows of the wave hold the same, valid voxel (kx, ky, kz are row-uniform values)
__device__ __forceinline__ bool rows_share_home(int kx, int ky, int kz) {
    const int ax = __builtin_amdgcn_readlane(kx, 0), ay = __builtin_amdgcn_readlane(ky, 0), az = __builtin_amdgcn_readlane(kz, 0);
    bool same = ax != INT_MIN;
#pragma unroll
    for (int l = 16; l < 64; l += 16)
        same = same && __builtin_amdgcn_readlane(kx, l) == ax && __builtin_amdgcn_readlane(ky, l) == ay &&
               __builtin_amdgcn_readlane(kz, l) == az;
    return same;
}

// NB = sweep half-width (1 -> 27 voxels, 2 -> 125); HIST = histogram-assisted selection.
// PROF: per-phase shader-clock accounting (s_memtime) summed over waves into prof[0..7]:
//   0 phase A (transform, voxel) | 1 hash probes + chunk lists | 2 candidate streaming | 3 in-stream prunes |
//   4 final selection | 5 covariance sums | 6 phase C (normal, residual, u) | 7 phase D (u u^T accumulation)
// SHARED: compile the shared-home-voxel path (NB = 1). Off in the default instantiation since round 2: with the carried-over bound
// and the slab-gated probes the generic path runs a B2 launch in 0.098 ms against 0.108 ms with the shared path (B2-small 0.071 / 0.075).
// The tile loop of the row search: the body of k_accumulate_rows, and — with `after_tile`, called by the whole wave when a tile's
// rounds are done, while W.id[] still names the tile's keypoints — of the persistent small-frame kernel (k_gn_persistent), which runs
// the residual part for the same keypoints right there.
template <int NB, bool HIST, bool PROF, bool SHARED, bool POOLS, bool STAGE = false, typename AfterTile>
__device__ __forceinline__ void rows_tiles(const MapView &map, const KpView &kp, const GnState *st, const GnParams &prm, const DebugView &dbg,
                                           int first_iter, int rounds, unsigned long long *prof, int ablate, char *smem, int tile_first, int tile_end,
                                           int tile_step, AfterTile after_tile) {
    constexpr int S = 2 * NB + 1, V = S * S * S, VIT = (V + 15) / 16, OCC = (V + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4, sub = lane & 15;
    WaveScratch<OCC> &W = reinterpret_cast<WaveScratch<OCC> *>(smem)[wave];
    RowList &R = W.list[row];
    RowProbe<OCC> &RP = W.probe[row];
    SharedStage &SH = W.stage;
    const int k = prm.max_nb;
    // pool: how many of a search's nearest candidates the keypoint's record keeps (the k neighbours + spare ones), see phase V
    const int pool_cap = (!POOLS || (ablate & 2048) || !kp.pools) ? k : min(KMAX, k + (((ablate >> 12) & 15) ? ((ablate >> 12) & 15) : POOL_EXTRA));
    const bool pool_on = pool_cap > k;
    // 27-voxel sweep, search without a carried-over bound (the first of a solve): after the first probe batch (the 16 nearest voxels) a row
    // that holds k candidates finds its k-th best, and the second batch's voxels (far edges and corners) are streamed only if their box
    // reaches inside it — what the 125-voxel sweep always does. On bounded searches the extra selection costs more than it saves (round 2).
    const bool cull1 = NB == 1 && !kp.kth_valid && (((ablate >> 18) & 1) != CTGN_CULL1_DEFAULT);
    const int blk = map.blk;
    const char *pbase = reinterpret_cast<const char *>(map.blocks);       // 32-bit byte offsets: host keeps blocks < 4 GiB
    const uint32_t stride3 = 3u * (uint32_t) blk * 8u;                     // bytes per block; uniform base + 32-bit lane offset loads

    unsigned long long pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // 6: pool checks; 8: keypoints certified by them, 9: search rounds, 10: hash probes issued, 11: map points fetched
    unsigned long long tprev = 0;
    if (PROF) tprev = __builtin_readcyclecounter();
    const unsigned long long t_wave_start = tprev;
#define CTGN_TICK(slot)                                                  \
    if (PROF) {                                                          \
        const unsigned long long now_ = __builtin_readcyclecounter();    \
        pc[slot] += now_ - tprev;                                        \
        tprev = now_;                                                    \
    }
    const int kp_per_wave = 4 * rounds;
    const int ntiles = (kp.n + kp_per_wave - 1) / kp_per_wave;
    // 125-voxel sweep, wave-shared probes (shared2): the home voxel whose sweep sits in W.socc and the per-axis offset sets (slab-mask
    // form) it was probed for; a later round with the same home voxel that needs no more than that takes the table as it is (round 5)
    int s2_hx = INT_MIN, s2_hy = 0, s2_hz = 0;
    uint32_t s2_mask = 0u;
    // STAGE: the home voxel whose inner neighbourhood sits in the wave's GroupStage, the slab-mask product it was filled for, its points;
    // g3_no_*: the last home voxel whose group did not fit the table (not tried again while the rounds stay there)
    GroupStage *const G3 = STAGE ? reinterpret_cast<GroupStage *>(smem + sizeof(WaveScratch<OCC>) * ROW_WAVES) + wave : nullptr;
    int g3_hx = INT_MIN, g3_hy = 0, g3_hz = 0, g3_P = 0;
    uint32_t g3_mask = 0u;
    int g3_no_hx = INT_MIN, g3_no_hy = 0, g3_no_hz = 0;
    int g3_rounds = 0, g3_fills = 0, g3_all_rounds = 0;     // STAGE statistics: rounds that streamed the table, table fills, search rounds
    int g3_over = 0, g3_points = 0, g3_eligible = 0;        // ... fills given up (more points than GCAP), points copied, rounds with an inner reach in one home voxel
    for (int tile = tile_first; tile < tile_end; tile += tile_step) {
        // ---------------- phase A: lane (row, sub < rounds) owns keypoint (sub * ntiles + tile) * 4 + row: round r of a
        // tile works on four CONSECUTIVE keypoints (neighbours in the scan usually share their home voxel), while the
        // rounds of one tile are spread over the whole scan with stride 4 * ntiles. Every tile is thereby a uniform
        // sample of the scan, which levels the per-wave work (dense and sparse regions differ ~4x in candidates per
        // keypoint; with contiguous tiles the slowest wave ran 1.8x the mean).
        // With kp.order (large scans over maps that exceed the caches, ctgn_api.hip) the positions are sorted by home voxel:
        // the four of a round nearly always share it, neighbouring tiles work on neighbouring voxels at the same time (L2
        // reuse). kp.chunk consecutive rounds take consecutive positions (the staged neighbourhood of one round then often
        // serves the next) and groups of kp.chunk rounds are strided over the whole order; the library passes 1 — longer
        // chunks unbalance the tiles by more than the reuse saves.
        int my_kp = -1, n_search = 0;
        bool compact = false;              // pools were checked: the keypoints still to be searched are compacted into rounds (W.slot)
        if (sub < rounds) {
            const int c = kp.chunk, g = sub / c, j = sub - g * c, cg = min(c, rounds - g * c);
            const int pos = g * ntiles * 4 * c + tile * 4 * cg + j * 4 + row;
            if (pos < kp.n) my_kp = kp.order ? (int) kp.order[pos] : pos;
        }
        const bool own = my_kp >= 0;
        {
        Vec3 p{0, 0, 0};
        int kxv = INT_MIN, kyv = 0, kzv = 0;
        float kbv = __int_as_float(0x7f800000), rr2v = 0.f;
        bool guessv = false;
        if (own && kp.guess2 > 0.f && !(kp.kth_valid && !first_iter) && !(ablate & 256)) { kbv = kp.guess2; guessv = true; }
        if (own) {
            const Vec3 raw{kp.rx[my_kp], kp.ry[my_kp], kp.rz[my_kp]};
            const double alpha = alpha_timestamp(kp.t[my_kp], st->tbe[0], st->tbe[1]);
            const Vec3 before{kp.wx[my_kp], kp.wy[my_kp], kp.wz[my_kp]};          // where the last iteration (or the upload) saw it
            if (first_iter) {
                p = before;
            } else {
                if (kp.resume) {
                    p = before;                                              // written by k_pool_check in this iteration
                } else {
                    p = ct_transform(st, alpha, raw);
                    kp.wx[my_kp] = p.x; kp.wy[my_kp] = p.y; kp.wz[my_kp] = p.z;
                }
                if (kp.kth_valid && !(ablate & 256)) {
                    const double dx = p.x - before.x, dy = p.y - before.y, dz = p.z - before.z;
                    const double moved = sqrt(sq_norm3(dx, dy, dz));
                    const double rprev = (double) kp.kth[2 * my_kp], kprev = (double) kp.kth[2 * my_kp + 1];
                    // kth[1] = distance of the previous k-th neighbour (rounded up). The k neighbours found then lie within that + the distance
                    // moved of the new position: an upper bound on the new k-th distance, and a search that admits only candidates inside
                    // it keeps ~k of the streamed points in its list (with fewer than k neighbours then: the radius, i.e. no bound).
                    if (kprev > 0.0) {
                        const double reach = kprev + moved;
                        kbv = __double2float_ru(reach * reach * (1.0 + 1e-9));
                    }
                    if (pool_on) {
                        // kth[0] = radius around the PREVIOUS position inside which every map point is in the keypoint's pool (rounded
                        // down; 0: no pool): around the new position that radius is smaller by the distance moved (the map does not
                        // change inside a solve)
                        const double rnow = rprev - moved * (1.0 + 1e-9) - 1e-12;
                        if (rnow > 0.0) rr2v = __double2float_rd(rnow * rnow * (1.0 - 1e-9));
                    }
                }
            }
            int a = voxel_coord(p.x, map.resolution), b = voxel_coord(p.y, map.resolution), c = voxel_coord(p.z, map.resolution);
            if (sweep_in_short_range(a, NB) && sweep_in_short_range(b, NB) && sweep_in_short_range(c, NB)) {
                kxv = a; kyv = b; kzv = c;
            } else {
                rr2v = 0.f;
            }
        }
        W.px[lane] = p.x; W.py[lane] = p.y; W.pz[lane] = p.z;
        W.kx[lane] = kxv; W.ky[lane] = kyv; W.kz[lane] = kzv;
        W.id[lane] = my_kp;
        W.kb[lane] = kbv;
        W.rr2[lane] = rr2v;
        W.todo[lane] = own ? 1 : 0;
        W.gs[lane] = guessv ? 1 : 0;
        }
        CTGN_TICK(0)
        // ---------------- phase V: pools. The previous bounded search (or pool check) of this solve left the keypoint a POOL — its nearest
        // candidates, the k neighbours among them, nearest first — and the radius inside which the pool is complete. If the new k-th
        // nearest pool member is still inside that radius (shrunk by the distance the keypoint has moved since), no map point outside the
        // pool can be among the k nearest, nor tie with one of them: the neighbour set is the k nearest pool members — up to KMAX
        // gathered points and one selection instead of the hash probes and the ~120 streamed candidates of a search. Exactly the same
        // set in the same order: the certificate is conservative (directed roundings), and a keypoint it does not cover takes the
        // search below, bounded by its pool.
        if (POOLS && pool_on && kp.kth_valid && !first_iter && !(ablate & 256) && any64(W.rr2[lane] > 0.f)) {
            compact = true;
            uint32_t *T = reinterpret_cast<uint32_t *>(RP.chunk);            // the row's pool: point byte offsets by pool index
            struct PoolRec { uint32_t hdr, o0, o1; };
            struct PoolPts { double x0, y0, z0, x1, y1, z1; };
            auto request = [&](int r, PoolRec &q) {                          // round r's record words
                const int id = W.id[row * 16 + min(r, rounds - 1)];
                const uint32_t *o = kp.sel + (size_t) max(id, 0) * SEL_STRIDE;
                q.hdr = o[0]; q.o0 = o[1 + sub]; q.o1 = o[17 + sub];
            };
            auto pool_size = [&](int r, const PoolRec &q) {                  // 0: the round's keypoint has no pool to check
                return (r < rounds && W.rr2[row * 16 + min(r, rounds - 1)] > 0.f) ? (int) ((q.hdr >> 8) & REC_N_MASK) : 0;
            };
            auto gather = [&](int m, const PoolRec &q, PoolPts &t) {         // its pool members, two per lane (masked lanes read offset 0)
                load_point(pbase, sub < m ? q.o0 : 0u, t.x0, t.y0, t.z0);
                load_point(pbase, sub + 16 < m ? q.o1 : 0u, t.x1, t.y1, t.z1);
            };
            // Three rounds in flight: while round r is selected, the points of round r + 1 and the record of round r + 2 are on their way
            PoolRec rec_cur{0u, 0u, 0u}, rec_nxt{0u, 0u, 0u}, rec_far{0u, 0u, 0u};
            PoolPts pts_cur{0, 0, 0, 0, 0, 0}, pts_nxt{0, 0, 0, 0, 0, 0};
            request(0, rec_cur);
            request(1, rec_nxt);
            gather(pool_size(0, rec_cur), rec_cur, pts_cur);
            for (int r = 0; r < rounds; ++r) {
                const int src = row * 16 + r;
                request(r + 2, rec_far);
                gather(pool_size(r + 1, rec_nxt), rec_nxt, pts_nxt);
                const float rr2 = W.rr2[src];
                const bool vrow = rr2 > 0.f;                                  // row-uniform
                if (any64(vrow)) {
                const int m = pool_size(r, rec_cur);
                const bool v0 = sub < m, v1 = sub + 16 < m;
                if (PROF) pc[11] += (unsigned long long) (__popcll(ballot64(v0)) + __popcll(ballot64(v1)));
                const double qx = W.px[src], qy = W.py[src], qz = W.pz[src];
                R.d2[sub] = sq_norm3(pts_cur.x0 - qx, pts_cur.y0 - qy, pts_cur.z0 - qz); R.vis[sub] = (uint32_t) sub;
                R.d2[sub + 16] = sq_norm3(pts_cur.x1 - qx, pts_cur.y1 - qy, pts_cur.z1 - qz); R.vis[sub + 16] = (uint32_t) (sub + 16);
                T[sub] = rec_cur.o0; T[sub + 16] = rec_cur.o1;
                bool tie = false;
                row_select<HIST>(R, m, KMAX, sub, row, map.r2thr, tie);     // the whole pool, sorted by its distances to the new position
                const double s0 = R.d2[sub], s1 = R.d2[sub + 16];
                const int n_in = row_sum_i32(((v0 && s0 <= map.r2thr) ? 1 : 0) + ((v1 && s1 <= map.r2thr) ? 1 : 0));   // map.h:491-493
                const int n = min(n_in, k);
                // needed: the k-th neighbour's distance — or, with fewer than k pool members inside the radius, the radius itself —
                // lies strictly inside the region the pool is known to be complete in
                const double need2 = n_in >= k ? R.d2[k - 1] : map.r2thr;
                const bool pass = vrow && need2 * (1.0 + 1e-8) < (double) rr2;
                const int kp_r = W.id[src];
                if (PROF) pc[8] += (unsigned long long) __popcll(ballot64(pass && sub == 0));
                if (PROF) {            // pc[7]: certified AND the n neighbours are the previous ones in the previous order
                    const bool same = (sub >= n || R.vis[sub] == (uint32_t) sub) && (sub + 16 >= n || R.vis[sub + 16] == (uint32_t) (sub + 16));
                    const bool keep = row_sum_i32(same ? 0 : 1) == 0 && (int) (rec_cur.hdr & REC_N_MASK) == n && !tie && !(rec_cur.hdr & TIE_FLAG);
                    pc[7] += (unsigned long long) __popcll(ballot64(pass && keep && sub == 0));
                }
                if (pass) {
                    uint32_t *o = kp.sel + (size_t) kp_r * SEL_STRIDE;
                    if (sub == 0) {
                        const uint32_t flagged = (uint32_t) n | ((uint32_t) m << 8) | (tie ? TIE_FLAG : 0u);
                        if (flagged != rec_cur.hdr) o[0] = flagged;
                        kp.cnt[kp_r] = flagged;
                        kp.kth[2 * kp_r] = __int_as_float(max(__float_as_int(sqrtf(rr2)) - 2, 0));          // rounded down
                        kp.kth[2 * kp_r + 1] = sqrt_bound_up(need2);
                        W.todo[src] = 0;
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int e = sub + 16 * h;
                        // nearest first: the n neighbours, then the rest of the pool — only the entries that moved (a settled keypoint's
                        // record is not written at all)
                        if (e < m && R.vis[e] != (uint32_t) e) o[1 + e] = T[R.vis[e]];
                    }
                } else if (vrow && sub == 0 && n_in >= k + POOL_REFILL) {
                    // searched below, admitting what lies within the (k + 4)-th pool member (a pool again, at the price of four candidates)
                    W.kb[src] = __double2float_ru(R.d2[k + POOL_REFILL - 1] * (1.0 + 0x1p-40));
                }
                }
                rec_cur = rec_nxt; rec_nxt = rec_far; pts_cur = pts_nxt;
            }
            CTGN_TICK(6)
        }

        // Two passes at most: pass 0 = the tile's rounds; pass 1 (only after a guessed bound, KpView::guess2) = the keypoints whose guess admitted
        // fewer than k candidates — or whose k-th lies within rounding of the guess —, compacted into rounds and searched on the radius.
#pragma nounroll
        for (int pass = 0; pass < CTGN_GUESS_PASSES; ++pass) {
        if (pass >= 1) {
            const bool again = W.todo[lane] == 2;
            if (!any64(again)) break;
            W.todo[lane] = again ? 1 : 0;
            // (CTGN_GUESS_PASSES 3: a keypoint its guess failed is searched within twice the guessed distance before it is searched on the radius)
            const float g4 = kp.guess2 * 4.f;
            const bool wider = CTGN_GUESS_PASSES > 2 && pass == 1 && (double) g4 < 0.8 * map.r2thr;
            if (again) { W.kb[lane] = wider ? g4 : __int_as_float(0x7f800000); W.gs[lane] = wider ? 1 : 0; }
            compact = true;
        }
        // ---------------- phase A2: the keypoints that are searched
        {
        const bool searched = W.todo[lane] != 0;
        // Per axis a voxel offset -1 / 0 / +1 is needed iff the slab of that offset lies within the keypoint's bound (offset 0 always):
        // the product of the three per-axis sets is a superset of the sweep voxels whose box reaches inside the bound — a superfluous
        // voxel only streams candidates that the bound then rejects. Once per keypoint, here, where every lane has its own.
        uint32_t mr = 0x1084u;                     // offset 0 of every axis
        const int kxv = W.kx[lane], kyv = W.ky[lane], kzv = W.kz[lane];
        if (searched && kxv != INT_MIN) {
            const double px = W.px[lane], py = W.py[lane], pz = W.pz[lane];
            const double bnd = fmin(map.r2thr, (double) W.kb[lane]) * (1.0 + 1e-8) + 1e-12;
#pragma unroll
            for (int o = -NB; o <= NB; ++o) {
                if (o == 0) continue;
                double g;
                g = axis_gap(px, kxv + o, map.resolution); mr |= g * g <= bnd ? 1u << (o + 2) : 0u;
                g = axis_gap(py, kyv + o, map.resolution); mr |= g * g <= bnd ? 1u << (5 + o + 2) : 0u;
                g = axis_gap(pz, kzv + o, map.resolution); mr |= g * g <= bnd ? 1u << (10 + o + 2) : 0u;
            }
        }
        W.mr[lane] = (uint16_t) ((ablate & 512) ? 0x7fffu : mr);
        // the searched keypoints, compacted into rounds of four: position in the order (sub, row), so that with nothing certified
        // (first search) round r holds the keypoints of lanes (0..3, r) as before — four consecutive ones
        const unsigned long long sm = ballot64(searched);
        int q = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) q += __popc(row_bits(sm, j) & ((1u << (sub + (j < row ? 1 : 0))) - 1u));
        n_search = (int) __popcll(sm);
        if (compact) {
            if (lane >= n_search) W.slot[lane] = 255;
            if (searched) W.slot[q] = (uint8_t) lane;
        }
        }

        // ---------------- phase B: the row works on the keypoint owned by its lane `r`
        Probe nxt;
        int nxt_v = 255, nxt_round = -1;            // generic path: probe batch already in flight for round nxt_round
        int st_kx = INT_MIN, st_ky = 0, st_kz = 0, st_P = 0;   // fast path: home voxel whose neighbourhood is staged, its size
        Probe snxt;                                 // fast path: the wave's probes for round snxt_round, already in flight
        int snxt_round = -1;
        unsigned long long need_next = 0, st_need = 0;    // bit i: the i-th nearest sweep voxel is needed by round need_round / is in the staged table
        int need_round = -1;
        // Which of the 27 sweep voxels of the shared home voxel can hold one of the k nearest of some row's keypoint of round rr: those in
        // the product of that keypoint's per-axis offset sets (W.mr, phase A; with a carried-over bound typically 1-4 voxels instead of 27).
        // Bit i = the i-th nearest sweep voxel, lane i's.
        // the lane whose keypoint row j works on in search round r (255: none). Without a pool check every keypoint of the tile is searched
        // and round r simply takes the lanes (0..3, r) — no table, no LDS round trip in front of the round's first loads.
        auto slot_of = [&](int r, int j) -> int { return compact ? (int) W.slot[4 * r + j] : j * 16 + r; };
        auto shared_need = [&](int rr) -> unsigned long long {
            const int svl = lane < 27 ? (int) c_sweep1.v[lane] : 13;
            const uint32_t want = (1u << (svl / 9 + 1)) | (1u << (5 + (svl / 3) % 3 + 1)) | (1u << (10 + svl % 3 + 1));    // offsets -1 .. +1 = bits 1 .. 3
            bool any_row = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sj = slot_of(rr, j);
                any_row = any_row || (sj != 255 && ((uint32_t) W.mr[sj & 63] & want) == want);
            }
            return ballot64(any_row && lane < 27);
        };
        const int search_rounds = compact ? (n_search + 3) >> 2 : rounds;
        for (int r = 0; r < ((ablate & 1024) ? 0 : search_rounds); ++r) {
            const int slot_r = slot_of(r, row);
            const bool idle = slot_r == 255;                       // this row has no keypoint in the last round
            const int src = idle ? row * 16 : slot_r;
            const double qx = W.px[src], qy = W.py[src], qz = W.pz[src];
            const int kx = idle ? INT_MIN : W.kx[src], ky = W.ky[src], kz = W.kz[src];
            const bool searching = kx != INT_MIN;
            const uint32_t lt_mask = (1u << sub) - 1u;
            int Ln = 0;
            // admission bound of the stream: the radius or the bound carried over from the previous search, then the k-th best so far
            double kth_d2 = fmin(map.r2thr, (double) W.kb[src]);
            // A search that starts with a bound (carried over from the previous one) keeps a pool: its bound exceeds the new k-th distance by
            // about the distance the keypoint has moved, so nearly everything it admits comes for free. A search bounded by the radius only
            // would have to carry the spare members through every cut of its long stream: it keeps the k neighbours and leaves no pool.
            const bool guessed = W.gs[src] != 0;         // (a guessed search keeps no pool: like the radius-only search it stands in for)
            // Round 5: ... and so does a search whose carried-over bound lies BEYOND the radius — previous k-th distance + distance moved, for a
            // far keypoint that the first correction of a solve moved by decimetres (or one in a sparse part of the map): it admits everything
            // within the radius, so the pool is complete out to the radius (or to its last member), and the keypoint is no longer searched on
            // the whole radius in every later iteration (B2: third search 0.080 -> 0.075 ms). Bit 29 of the ablation mask: as before.
            const bool carried_beyond = !guessed && !((double) W.kb[src] < map.r2thr) && W.kb[src] < __int_as_float(0x7f800000) && !(ablate & (1 << 29));
            const int kpool = (!guessed && ((double) W.kb[src] < map.r2thr || carried_beyond)) ? pool_cap : k;

            // Do the four keypoints of this round live in the same home voxel? (wave-uniform test on SGPRs)
            const bool uniform_home = SHARED && (NB == 1 && blk <= 32) && !(ablate & 32) && rows_share_home(kx, ky, kz);
            const uint32_t *occ_tab = RP.occ;     // where B4 finds a candidate's voxel block
            bool tie_seen = false;                // some selection of this round met candidates whose distances (nearly) tie
            if (PROF) pc[9] += 1;
            if (STAGE) ++g3_all_rounds;

            if (uniform_home) {
                // ===== fast path: shared, flattened neighbourhood =====
                occ_tab = SH.occ;
                const unsigned long long need = (need_round == r) ? need_next : shared_need(r);
                if (!(kx == st_kx && ky == st_ky && kz == st_kz && (need & ~st_need) == 0ull)) {
                    // probe the sweep voxels some row needs, once for the wave: lane i < 27 takes the i-th NEAREST sweep voxel (centre,
                    // faces, edges, corners), so the flattened table starts with the home voxel's points and the first prune of the
                    // stream already leaves a tight k-th-best bound for the rest
                    const int sv = lane < 27 ? (int) c_sweep1.v[lane] : 27;
                    if (snxt_round != r)
                        snxt = probe_issue(map, ((need >> lane) & 1ull) && !(ablate & 16), kx + sv / 9 - 1, ky + (sv / 3) % 3 - 1, kz + sv % 3 - 1);
                    st_need = need;
                    if (PROF) pc[10] += (unsigned long long) __popcll(ballot64(snxt.active));
                    const uint32_t bc = probe_resolve(map, snxt);
                    const int cnt = (int) (bc & 127u);
                    const int inc = row_scan_i32(cnt);                       // inclusive prefix within each DPP row
                    const int tot0 = __builtin_amdgcn_readlane(inc, 15), tot1 = __builtin_amdgcn_readlane(inc, 31);
                    const int pre = inc - cnt + (row == 1 ? tot0 : 0);       // flat position of this voxel's first point
                    if (lane < 28) SH.occ[sv] = bc;
                    int nchunk = 0;
                    for (int hh = 0; hh < 2; ++hh) {
                        const int left = cnt - 16 * hh;
                        const bool has = left > 0;
                        const unsigned long long hb = ballot64(has);
                        if (!hb) break;
                        if (has) SH.chunk[nchunk + __popcll(hb & ((1ull << lane) - 1ull))] =
                                make_uint2((bc >> 7) * stride3 + 16u * POINT_BYTES * hh,
                                           ((((uint32_t) sv << 6) | (16u * hh)) << 15) | ((uint32_t) (pre + 16 * hh) << 5) | (uint32_t) min(left, 16));
                        nchunk += __popcll(hb);
                    }
                    // flatten: 4 chunks per step, one per row; candidate c of the table = (offset of its x, visit index)
                    for (int c0 = 0; c0 < nchunk; c0 += 4) {
                        const int c = c0 + row;
                        if (c < nchunk) {
                            const uint2 ch = SH.chunk[c];
                            const uint32_t n16 = ch.y & 31u, dest = (ch.y >> 5) & 1023u, visb = ch.y >> 15;
                            if ((uint32_t) sub < n16) {
                                SH.vis[dest + sub] = (uint16_t) (visb + (uint32_t) sub);
                            }
                        }
                    }
                    st_kx = kx; st_ky = ky; st_kz = kz;
                    st_P = tot0 + tot1;
                }
                CTGN_TICK(1)
                // The four rows hold four DIFFERENT queries over the SAME candidate table: every lane fetches its own candidate (64
                // distinct ones per step instead of 16 fetched four times) and tests it against all four queries, whose coordinates
                // and admission bounds are wave-uniform (SGPRs). The chain of dependent load round trips — what this phase waits
                // on — is a quarter as long; each row's list receives the same candidates in the same order as before.
                // Two register sets in ping-pong: the loads of the next 64 candidates are in flight while the current 64 are tested.
                auto bcast = [](double v, int l) {
                    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
                };
                double Qx[4], Qy[4], Qz[4], Kth[4];
                int L[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Qx[j] = bcast(qx, 16 * j); Qy[j] = bcast(qy, 16 * j); Qz[j] = bcast(qz, 16 * j);
                    Kth[j] = bcast(kth_d2, 16 * j);
                    L[j] = 0;
                }
                const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
                struct Cand { double x, y, z; uint32_t vis; bool valid; };
                auto fetch = [&](int c0, Cand &o) {
                    const int c = c0 + lane;
                    o.valid = c < st_P;
                    const int cc = o.valid ? c : 0;
                    o.vis = SH.vis[cc];
                    // the candidate's byte offset from its visit index: block of its sweep voxel + slot (no offset table in LDS: the
                    // 3.4 KB it took per wave are what kept the block under the CU's LDS at three blocks)
                    const uint32_t off = o.valid ? (SH.occ[o.vis >> 6] >> 7) * stride3 + (o.vis & 63u) * POINT_BYTES : 0u;   // masked lanes read the start of the block storage
                    load_point(pbase, off, o.x, o.y, o.z);
                };
                auto test = [&](const Cand &cnd) {
                    if (PROF) pc[11] += 4ull * (unsigned long long) __popcll(ballot64(cnd.valid));   // per keypoint, as the row path counts
                    double d2[4];
                    unsigned long long pm[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double dx = cnd.x - Qx[j], dy = cnd.y - Qy[j], dz = cnd.z - Qz[j];
                        d2[j] = sq_norm3(dx, dy, dz);
                        // the stream filter is only an optimisation: a candidate that ties the current k-th best in d2 is
                        // admitted whatever its visit index, and the exact total order (d2, vis) decides in row_select
                        pm[j] = ballot64(cnd.valid && d2[j] <= Kth[j] && !(ablate & 128));
                    }
                    // a list that cannot take this step's admissions: cut every row back to its k best first, then admit against the
                    // tighter bounds (after the cut a list holds <= k <= 32 entries and a step adds <= 64)
                    if (max(max(L[0] + (int) __popcll(pm[0]), L[1] + (int) __popcll(pm[1])),
                            max(L[2] + (int) __popcll(pm[2]), L[3] + (int) __popcll(pm[3]))) > LCAP) {
                        CTGN_TICK(2)
                        Ln = row == 0 ? L[0] : row == 1 ? L[1] : row == 2 ? L[2] : L[3];
                        Ln = row_select<HIST>(R, Ln, kpool, sub, row, kth_d2, tie_seen);
                        if (Ln >= kpool) kth_d2 = kth_bound(R.d2[kpool - 1], map.r2thr);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            L[j] = __builtin_amdgcn_readlane(Ln, 16 * j);
                            Kth[j] = bcast(kth_d2, 16 * j);
                            pm[j] = ballot64(cnd.valid && d2[j] <= Kth[j]);
                        }
                        CTGN_TICK(3)
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if ((pm[j] >> lane) & 1ull) {
                            const int pos = L[j] + __popcll(pm[j] & below);
                            W.list[j].d2[pos] = d2[j];
                            W.list[j].vis[pos] = cnd.vis;
                        }
                        L[j] += __popcll(pm[j]);
                    }
                };
                // straight-line loop body: the fetches are unconditional (out-of-range lanes re-read candidate 0 and are
                // masked), so the compiler's s_waitcnt covers exactly the set being consumed, not the one in flight
                Cand ca{0, 0, 0, 0, false}, cb{0, 0, 0, 0, false};
                fetch(0, ca);
                for (int c0 = 0; c0 < ((ablate & 1) ? 0 : st_P); c0 += 128) {
                    fetch(c0 + 64, cb);
                    test(ca);
                    fetch(c0 + 128, ca);
                    test(cb);
                }
                Ln = row == 0 ? L[0] : row == 1 ? L[1] : row == 2 ? L[2] : L[3];
                CTGN_TICK(2)
            } else {
            // ===== generic path: every row probes and streams its own keypoint's neighbourhood =====
            st_kx = INT_MIN;                      // the shared stage aliases the per-row probe scratch
            const double r2bound = kth_d2;        // the probes are culled against the bound the round starts with (27-voxel sweep: both batches)
            const uint32_t mreach = W.mr[src];
            // 125-voxel sweep, the (active) keypoints of the round in ONE home voxel — the rule on an upload sorted by home voxel (config D):
            // the 125 hash probes are issued once for the wave, two per lane, all in flight together, instead of eight dependent batches of
            // 16 per row (into a table far larger than the caches a first search spends 44 % of its wave time there); the rows then take
            // their sweep voxels from the wave's table, each with its own reach test against its own current bound, as before.
            bool shared2 = false;
            bool staged3 = false;                 // STAGE: this round streams the wave's GroupStage
            const bool share_on = ((ablate >> 22) & 1) != CTGN_SHARED2_DEFAULT_ON;
            if (NB == 2 && (STAGE || share_on)) {
                int hx = INT_MIN, hy = 0, hz = 0;
                bool same = true;
#pragma unroll
                for (int l = 0; l < 64; l += 16) {
                    const int x = __builtin_amdgcn_readlane(kx, l), y = __builtin_amdgcn_readlane(ky, l), z = __builtin_amdgcn_readlane(kz, l);
                    if (x != INT_MIN) {
                        if (hx == INT_MIN) { hx = x; hy = y; hz = z; }
                        else same = same && x == hx && y == hy && z == hz;
                    }
                }
                shared2 = same && hx != INT_MIN;
                if (shared2) {
                    // which sweep voxels some row can reach (slab masks only: a superset of the rows' exact tests). The wave probes the
                    // PRODUCT of the per-axis unions of the four rows' offset sets — a superset of the union of their products that one
                    // 15-bit mask describes, so that a later round of the same home voxel can tell whether the table already holds
                    // what it needs (consecutive rounds of a chunked tile, KpView::chunk > 1, share their home voxel as a rule).
                    const uint32_t m0 = (uint32_t) __builtin_amdgcn_readlane((int) (searching ? mreach : 0u), 0);
                    const uint32_t m1 = (uint32_t) __builtin_amdgcn_readlane((int) (searching ? mreach : 0u), 16);
                    const uint32_t m2 = (uint32_t) __builtin_amdgcn_readlane((int) (searching ? mreach : 0u), 32);
                    const uint32_t m3 = (uint32_t) __builtin_amdgcn_readlane((int) (searching ? mreach : 0u), 48);
                    const uint32_t mu = m0 | m1 | m2 | m3;
                    // the wave's probes of the home voxel's sweep for the slab-mask product `mask`, two voxels per lane, into W.socc
                    auto probe_home = [&](uint32_t mask) {
                        auto wanted = [&](int v) {
                            const int ox = v / (S * S), oy = (v / S) % S, oz = v % S;            // offsets + NB: bit positions 0 .. 4
                            return v < V && ((mask >> ox) & (mask >> (5 + oy)) & (mask >> (10 + oz)) & 1u) != 0u;
                        };
                        const int v0 = lane, v1 = lane + 64;
                        const bool w0 = wanted(v0) && !(ablate & 16), w1 = wanted(v1) && !(ablate & 16);
                        Probe p0 = probe_issue(map, w0, hx + v0 / (S * S) - NB, hy + (v0 / S) % S - NB, hz + v0 % S - NB);
                        Probe p1 = probe_issue(map, w1, hx + (v1 % V) / (S * S) - NB, hy + ((v1 % V) / S) % S - NB, hz + (v1 % V) % S - NB);
                        if (PROF) pc[10] += (unsigned long long) (__popcll(ballot64(w0)) + __popcll(ballot64(w1)));
                        const uint32_t b0 = probe_resolve(map, p0), b1 = probe_resolve(map, p1);
                        W.socc[v0] = b0;
                        W.socc[v1] = b1;
                        s2_hx = hx; s2_hy = hy; s2_hz = hz; s2_mask = mask;
                    };
                    if constexpr (STAGE) {
                        // no row of the round reaches beyond the voxels next to the home voxel (no +-2 bit in any slab mask): the round can
                        // stream the group's table — the one in LDS if it covers the round's mask, else it is filled now
                        if ((mu & 0x4631u) == 0u && !(ablate & (1 << 28))) {
                            ++g3_eligible;
                            bool have = hx == g3_hx && hy == g3_hy && hz == g3_hz && (mu & ~g3_mask) == 0u;
                            if (!have && !(hx == g3_no_hx && hy == g3_no_hy && hz == g3_no_hz)) {
                                // the mask of the GROUP: every keypoint of the tile that is still to be searched in this home voxel with an
                                // inner reach (W.mr, phase A2) — with chunked tiles (KpView::chunk) those are the following rounds
                                const bool mine = W.todo[lane] == 1 && W.kx[lane] == hx && W.ky[lane] == hy && W.kz[lane] == hz &&
                                                  ((uint32_t) W.mr[lane] & 0x4631u) == 0u;
                                uint32_t mg = mine ? (uint32_t) W.mr[lane] : 0u;
                                mg |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) mg, 0xB1, 0xf, 0xf, false);
                                mg |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) mg, 0x4E, 0xf, 0xf, false);
                                mg |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) mg, 0x141, 0xf, 0xf, false);
                                mg |= (uint32_t) __builtin_amdgcn_update_dpp(0, (int) mg, 0x140, 0xf, 0xf, false);
                                mg = mu | (uint32_t) (__builtin_amdgcn_readlane((int) mg, 0) | __builtin_amdgcn_readlane((int) mg, 16) |
                                                      __builtin_amdgcn_readlane((int) mg, 32) | __builtin_amdgcn_readlane((int) mg, 48));
                                auto stage_group = [&](uint32_t mask) -> bool {
                                    probe_home(mask);
                                    GroupStage &G = *G3;
                                    // lane i < 27 takes the i-th NEAREST inner voxel (the first 27 entries of the nearest-first order are the
                                    // +-1 cube): the table starts with the home voxel's points
                                    const int sv = lane < 27 ? (int) c_sweep2.v[lane] : 0;
                                    const uint32_t bc = lane < 27 ? W.socc[sv] : 0u;
                                    const int cnt = (int) (bc & 127u);
                                    const int inc = row_scan_i32(cnt);
                                    const int tot0 = __builtin_amdgcn_readlane(inc, 15), tot1 = __builtin_amdgcn_readlane(inc, 31);
                                    g3_hx = INT_MIN;
                                    if (tot0 + tot1 > GCAP) { ++g3_over; return false; }
                                    const int pre = inc - cnt + (row == 1 ? tot0 : 0);           // flat position of this voxel's first point
                                    if (lane < 28) G.vinfo[lane] = make_uint2((bc >> 7) * stride3, ((uint32_t) sv << 20) | ((uint32_t) pre << 8) | (uint32_t) cnt);
                                    // copy: 16 points of four voxels per step (one voxel per row), every load of a pass in flight together
                                    for (int hh = 0; hh < 4; ++hh) {
                                        if (!ballot64(cnt > 16 * hh)) break;
                                        double cx[7], cy[7], cz[7];
                                        uint2 vi[7];
#pragma unroll
                                        for (int g = 0; g < 7; ++g) {
                                            vi[g] = G.vinfo[4 * g + row];
                                            const bool ok = sub < (int) (vi[g].y & 127u) - 16 * hh;
                                            load_point(pbase, vi[g].x + (ok ? (uint32_t) (16 * hh + sub) * POINT_BYTES : 0u), cx[g], cy[g], cz[g]);
                                        }
#pragma unroll
                                        for (int g = 0; g < 7; ++g) {
                                            if (sub < (int) (vi[g].y & 127u) - 16 * hh) {
                                                const int d = (int) ((vi[g].y >> 8) & 0xfffu) + 16 * hh + sub;
                                                G.x[d] = cx[g]; G.y[d] = cy[g]; G.z[d] = cz[g];
                                                G.vis[d] = (uint16_t) (((vi[g].y >> 20) << 6) | (uint32_t) (16 * hh + sub));
                                            }
                                        }
                                    }
                                    if (PROF) pc[11] += (unsigned long long) (tot0 + tot1);
                                    g3_hx = hx; g3_hy = hy; g3_hz = hz; g3_mask = mask; g3_P = tot0 + tot1;
                                    ++g3_fills;
                                    g3_points += tot0 + tot1;
                                    return true;
                                };
                                have = stage_group(mg) || (mg != mu && stage_group(mu));
                                if (!have) { g3_no_hx = hx; g3_no_hy = hy; g3_no_hz = hz; }
                            }
                            staged3 = have;
                        }
                    }
                    if (!staged3 && !share_on) {
                        shared2 = false;          // STAGE without the wave-shared probes: a round the table does not serve probes per row
                    } else {
                        const bool staged = staged3 || (!(ablate & (1 << 27)) && hx == s2_hx && hy == s2_hy && hz == s2_hz && (mu & ~s2_mask) == 0u);
                        if (!staged) probe_home(mu);
                    }
                }
            }
            if (STAGE && staged3) {
                // ===== the round streams the group's table: 16 candidates per row and step, the same ones for the four rows (LDS broadcast
                // reads), each tested against its row's own query and bound; lists, cuts and everything behind them as on the generic path
                occ_tab = W.socc;
                const GroupStage &G = *G3;
                const int P = g3_P;
                ++g3_rounds;
                CTGN_TICK(1)
                auto test3 = [&](double x, double y, double z, uint32_t vis, bool valid) {
                    const double dx = x - qx, dy = y - qy, dz = z - qz;
                    const double d2 = sq_norm3(dx, dy, dz);
                    const bool pass = valid && d2 <= kth_d2;
                    const uint32_t pm = row_bits(ballot64(pass), row);
                    if (pass) {
                        const int pos = Ln + __popc(pm & lt_mask);
                        R.d2[pos] = d2;
                        R.vis[pos] = vis;
                    }
                    Ln += __popc(pm);
                };
                for (int s0 = 0; s0 < P; s0 += 32) {
                    const int ia = s0 + sub, ib = ia + 16;
                    const int ca = min(ia, GCAP - 1), cb = min(ib, GCAP - 1);
                    const double xa = G.x[ca], ya = G.y[ca], za = G.z[ca];
                    const uint32_t va = G.vis[ca];
                    const double xb = G.x[cb], yb = G.y[cb], zb = G.z[cb];
                    const uint32_t vb = G.vis[cb];
                    if (PROF) pc[11] += (unsigned long long) min(32, P - s0);
                    test3(xa, ya, za, va, searching && ia < P);
                    test3(xb, yb, zb, vb, searching && ib < P);
                    if (any64(Ln > LCAP - 32)) {
                        CTGN_TICK(2)
                        Ln = row_select<HIST>(R, Ln, kpool, sub, row, kth_d2, tie_seen);
                        if (Ln >= kpool) kth_d2 = kth_bound(R.d2[kpool - 1], map.r2thr);
                        CTGN_TICK(3)
                    }
                }
                CTGN_TICK(2)
            } else {
            if (!shared2 && nxt_round != r) nxt = issue_batch<NB>(map, 0, sub, searching, kx, ky, kz, qx, qy, qz, nxt_v, ablate, r2bound, mreach);
            // B1 + B2, interleaved per batch of 16 sweep voxels (nearest voxels first):
            //   probe 16 voxels (one per lane) -> RP.occ[v] -> one chunk per 16 points of each occupied voxel -> the row
            //   streams the chunks (a voxel's x | y | z runs are contiguous, so a chunk is three 128-byte reads), with the
            //   loads of chunk c+1 in flight while chunk c is tested against the radius / current k-th best and
            //   compacted into the row's LDS candidate list. A candidate's visit index is (sweep index v << 6) | slot:
            //   the reference's x-major sweep + insertion order (map.h:470-480), whatever the probing order.
            // 125-voxel sweep: the voxels next to the home voxel (every offset within +-1) are the first 27 entries of the nearest-first order,
            // i.e. probe batches 0 and 1; when no searching row of the round reaches farther on any axis (its slab mask has no +-2 bit: every
            // bounded or guessed search on a level whose voxels are wider than the bound), batches 2 .. 7 hold nothing any row would accept
            // and are not run at all (they used to cost a reach test per lane and batch). Bit 26 of the ablation mask: off (A/B).
            int vit_run = VIT;
            if constexpr (NB == 2) {
                if (!(ablate & (1 << 26)) && !any64(searching && (mreach & 0x4631u) != 0u)) vit_run = 2;
            }
CTGN_BATCH_UNROLL
            for (int it = 0; it < VIT; ++it) {
                if (NB == 2 && it >= vit_run) continue;      // (not `break`: the loop stays unrolled)
                // the probe batch issued one step earlier is consumed now; the next batch (same keypoint, or the first
                // batch of the next round's keypoint) is issued before the chunk streaming so its latency is covered
                Probe cur = nxt;
                int cur_v = nxt_v;
                uint32_t bc_shared = 0u;
                if (shared2) {
                    // the row's own reach test for its lane's voxel of this batch (slab mask, then the exact box-to-sphere test against
                    // the row's CURRENT bound), the voxel's slot from the wave's table
                    int vx_, vy_, vz_;
                    cur.active = batch_reach<NB>(map, it, sub, searching, kx, ky, kz, qx, qy, qz, cur_v, vx_, vy_, vz_, ablate, fmin(r2bound, kth_d2), mreach);
                    bc_shared = cur.active ? W.socc[cur_v == 255 ? 0 : cur_v] : 0u;
                } else
                if (it + 1 < vit_run) {
                    // (against the row's CURRENT bound: on the 125-voxel sweep a first search knows its k-th best after the first batch or
                    // two, and the remaining batches then probe only the voxels that bound can still reach instead of every voxel
                    // within the radius — hash probes into a map larger than the caches are what that sweep waits for)
                    nxt = issue_batch<NB>(map, it + 1, sub, searching, kx, ky, kz, qx, qy, qz, nxt_v, ablate, NB == 2 ? fmin(r2bound, kth_d2) : r2bound, mreach);
                } else if (r + 1 < search_rounds) {
                    const int slot2 = slot_of(r + 1, row);
                    const int src2 = slot2 == 255 ? row * 16 : slot2;
                    const int kx2 = slot2 == 255 ? INT_MIN : W.kx[src2];
                    nxt = issue_batch<NB>(map, 0, sub, kx2 != INT_MIN, kx2, W.ky[src2], W.kz[src2], W.px[src2], W.py[src2], W.pz[src2], nxt_v, ablate,
                                          fmin(map.r2thr, (double) W.kb[src2]), (uint32_t) W.mr[src2]);
                    nxt_round = r + 1;
                }
                if (PROF && !shared2) pc[10] += (unsigned long long) __popcll(ballot64(cur.active));
                // a batch none of whose voxels any row can reach (most batches of a bounded 125-voxel sweep): nothing to resolve or stream
                if (any64(cur.active)) {
                const uint32_t bc = shared2 ? bc_shared : probe_resolve(map, cur);
                if (bc) RP.occ[cur_v] = bc;
                // Once the row holds k candidates and knows its k-th best distance, a voxel that lies entirely farther away cannot
                // contribute (its points would fail `d2 <= kth_d2` one by one): skip its chunks. The sweep goes nearest voxels
                // first, so on a dense map most of the later voxels fall here. Same slack as the radius cull. Only for the
                // 125-voxel sweep (8 probe batches): on the 27-voxel sweep (2 batches) the extra selection costs more than the
                // culled chunks save (B2: 0.094 -> 0.105 ms per launch), on the 125-voxel one it wins (D: 3.11 -> 2.31 ms).
                bool within_kth = true;
                if ((NB == 2 || cull1) && it > 0 && kth_d2 < map.r2thr) {
                    const int vv = (cur_v == 255) ? 0 : cur_v;
                    const double gx = axis_gap(qx, kx + vv / (S * S) - NB, map.resolution), gy = axis_gap(qy, ky + (vv / S) % S - NB, map.resolution),
                                 gz = axis_gap(qz, kz + vv % S - NB, map.resolution);
                    within_kth = gx * gx + gy * gy + gz * gz <= kth_d2 * (1.0 + 1e-8) + 1e-12;
                }
                const int cnt_mine = within_kth ? (int) (bc & 127u) : 0;
                const uint32_t off_mine = (bc >> 7) * stride3;
                int nchunk = 0;
                for (int hh = 0; hh < 4; ++hh) {
                    const int left = cnt_mine - 16 * hh;
                    const bool has = left > 0;
                    const unsigned long long hb = ballot64(has);
                    if (!hb) break;
                    const uint32_t hm = row_bits(hb, row);
                    if (has) RP.chunk[nchunk + __popc(hm & lt_mask)] =
                            make_uint2(off_mine + 16u * POINT_BYTES * hh, ((((uint32_t) cur_v << 6) | (16u * hh)) << 8) | (uint32_t) min(left, 16));
                    nchunk += __popc(hm);
                }
                CTGN_TICK(1)
                struct Cand { double x, y, z; uint32_t vis; bool valid; };
                auto fetch = [&](int c, Cand &o) {
                    // unconditional (straight-line) fetch: past the end of the row's chunk list it reads offset 0 with
                    // zero valid points, so no control flow sits between the loads and the waits that cover them
                    uint2 ch = RP.chunk[c & 63];
                    if (c >= nchunk) ch = make_uint2(0u, 0u);
                    o.valid = (uint32_t) sub < (ch.y & 0xffu);
                    const uint32_t off = ch.x + (o.valid ? (uint32_t) sub * POINT_BYTES : 0u);          // legal address either way
                    load_point(pbase, off, o.x, o.y, o.z);                                              // x, y: one 16-byte load; z: one 8-byte load
                    o.vis = (ch.y >> 8) + (uint32_t) sub;
                };
                auto test = [&](const Cand &cnd) {
                    if (PROF) pc[11] += (unsigned long long) __popcll(ballot64(cnd.valid));
                    const double dx = cnd.x - qx, dy = cnd.y - qy, dz = cnd.z - qz;
                    const double d2 = sq_norm3(dx, dy, dz);
                    const bool pass = cnd.valid && d2 <= kth_d2 && !(ablate & 128);
                    const uint32_t pm = row_bits(ballot64(pass), row);
                    if (pass) {
                        const int pos = Ln + __popc(pm & lt_mask);
                        R.d2[pos] = d2;
                        R.vis[pos] = cnd.vis;
                    }
                    Ln += __popc(pm);
                };
#if CTGN_STREAM_DEPTH == 3
                // three register sets: two chunks' loads are in flight while the third is tested (a list takes three chunks' admissions
                // between two capacity checks: 48 entries)
                Cand ca{0, 0, 0, 0, false}, cb{0, 0, 0, 0, false}, cc{0, 0, 0, 0, false};
                fetch(0, ca);
                fetch(1, cb);
                for (int c = 0; !(ablate & 1) && any64(c < nchunk); c += 3) {
                    fetch(c + 2, cc);
                    CTGN_STREAM_FENCE
                    test(ca);
                    fetch(c + 3, ca);
                    CTGN_STREAM_FENCE
                    test(cb);
                    fetch(c + 4, cb);
                    CTGN_STREAM_FENCE
                    test(cc);
                    if (any64(Ln > LCAP - 48)) {
                        CTGN_TICK(2)
                        Ln = row_select<HIST>(R, Ln, kpool, sub, row, kth_d2, tie_seen);
                        if (Ln >= kpool) kth_d2 = kth_bound(R.d2[kpool - 1], map.r2thr);
                        CTGN_TICK(3)
                    }
                }
#else
                Cand ca{0, 0, 0, 0, false}, cb{0, 0, 0, 0, false};
                fetch(0, ca);
                for (int c = 0; !(ablate & 1) && any64(c < nchunk); c += 2) {
                    fetch(c + 1, cb);
                    CTGN_STREAM_FENCE
                    test(ca);
                    fetch(c + 2, ca);
                    CTGN_STREAM_FENCE
                    test(cb);
                    if (any64(Ln > LCAP - 32)) {
                        // list nearly full somewhere in the wave: cut every row back to its k best
                        CTGN_TICK(2)
                        Ln = row_select<HIST>(R, Ln, kpool, sub, row, kth_d2, tie_seen);
                        if (Ln >= kpool) kth_d2 = kth_bound(R.d2[kpool - 1], map.r2thr);
                        CTGN_TICK(3)
                    }
                }
#endif
                }
                CTGN_TICK(2)
                // a row that has k candidates but no bound yet: find its k-th best now, so that the remaining (farther) voxels
                // of the sweep can be culled against it (refreshing the bound after every batch that added candidates was
                // measured too: D 2.31 -> 2.40-2.43 ms, the selections cost more than the tighter bound saves)
                if ((NB == 2 || cull1) && it + 1 < vit_run && any64(Ln >= kpool && !(kth_d2 < map.r2thr))) {
                    Ln = row_select<HIST>(R, Ln, kpool, sub, row, kth_d2, tie_seen);
                    if (Ln >= kpool) kth_d2 = kth_bound(R.d2[kpool - 1], map.r2thr);
                    CTGN_TICK(3)
                }
            }
            }
            }
            // the next round's shared probes go out now, so their latency hides behind this round's selection and sums
            if (SHARED && NB == 1 && blk <= 32 && r + 1 < search_rounds) {
                const int slot2 = slot_of(r + 1, row);
                const int src2 = slot2 == 255 ? row * 16 : slot2;
                const int kx2 = slot2 == 255 ? INT_MIN : W.kx[src2], ky2 = W.ky[src2], kz2 = W.kz[src2];
                if (!(ablate & 32) && rows_share_home(kx2, ky2, kz2)) {
                    const unsigned long long need2 = shared_need(r + 1);
                    need_next = need2;
                    need_round = r + 1;
                    if (!(kx2 == st_kx && ky2 == st_ky && kz2 == st_kz && (need2 & ~st_need) == 0ull)) {     // not served by the staged table
                        const int sv2 = lane < 27 ? (int) c_sweep1.v[lane] : 27;
                        snxt = probe_issue(map, ((need2 >> lane) & 1ull) && !(ablate & 16), kx2 + sv2 / 9 - 1, ky2 + (sv2 / 3) % 3 - 1,
                                           kz2 + sv2 % 3 - 1);
                        snxt_round = r + 1;
                    }
                }
            }
            // B3: final selection -> list sorted ascending, [0..n). A row that ends with fewer than min_number_neighbors (or 5)
            // candidates was never pruned (Ln < k throughout), so Ln already is its exact neighbour count, and the residual kernel
            // drops it on that count alone: when no row of the wave can be used, skip the selection and hand over counts only.
            // With pools every row is sorted and kept, the empty ones too: a keypoint with too few neighbours to be used (or none within
            // reach) is then certified as such by phase V instead of being searched again in every iteration.
            const bool row_needed = (Ln >= prm.min_nb && Ln >= 5) || dbg.n_nb != nullptr || (pool_on && searching);
            const int admitted = Ln;
            if (!(ablate & 2) && any64(row_needed)) Ln = row_select<HIST>(R, Ln, kpool, sub, row, kth_d2, tie_seen);
            const int m = min(Ln, kpool);          // pool: what the record keeps, sorted nearest first
            const int n = min(Ln, k);              // the neighbours (every list entry lies within the radius, map.h:491-493)
            // a guessed bound proves its result only if it admitted k candidates with the k-th clear of the guess: anything else is
            // searched again on the radius (pass 1) and hands nothing over now
            bool retry = false;
            if (guessed && searching) {
                // R.d2[k - 1] is the k-th smallest only behind a selection: a row that was not selected (min_number_neighbors above k, or the
                // selection ablated) is searched again rather than trusted
                const bool selected = row_needed && !(ablate & 2);
                retry = admitted < k || !selected || !(R.d2[k - 1] * (1.0 + 0x1p-48) < (double) W.kb[src]);
                if (retry && sub == 0) W.todo[src] = 2;
            }
            CTGN_TICK(4)
            // B4: hand the keypoint's neighbour set over: the block-storage byte offsets of the n kept points, nearest first
            // (the reference's neighbour vector is the same set farthest first: its readers walk the record backwards), in a
            // per-keypoint record; behind them the rest of the pool (phase V of the next iteration). The covariance sums are then
            // taken by the residual kernel — no cross-lane reductions and no point loads here.
            {
                const int kp_r = (idle || retry) ? -1 : W.id[src];
                if (kp_r >= 0 && !(ablate & 4)) {
                    uint32_t *o = kp.sel + (size_t) kp_r * SEL_STRIDE;
                    const bool sorted = row_needed && !(ablate & 2);
                    if (sub == 0) {
                        // (nearly) tied candidates: which of them the reference keeps, and in which order, only its own queue says
                        // (map.h:491-513) — the kernels that read this record replay it for this keypoint (resolve_ties)
                        const uint32_t flagged = (uint32_t) n | ((uint32_t) (sorted ? m : 0) << 8) | ((tie_seen && row_needed) ? TIE_FLAG : 0u);
                        o[0] = flagged; kp.cnt[kp_r] = flagged;
                        if (kp.kth) {
                            float kthv = 0.f;
                            if (sorted && kpool > k && !(carried_beyond && n < k)) {      // (fewer than k within the radius: no check could ever pass)
                                // Everything this search did not put into the list lies farther than the admission bound it ended with
                                // (culled voxels and rejected candidates were beyond the bound of their moment, and bounds only shrink);
                                // everything the selections dropped lies at or beyond the last pool member. Inside the smaller of the two
                                // the pool is complete.
                                const double r2 = admitted > kpool ? fmin(kth_d2, R.d2[kpool - 1]) : kth_d2;
                                kthv = sqrt_bound_down(r2);
                            }
                            kp.kth[2 * kp_r] = kthv;
                            kp.kth[2 * kp_r + 1] = sqrt_bound_up(sorted && n >= k ? R.d2[k - 1] : map.r2thr);
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int e = sub + 16 * h;
                        if (e < m && row_needed) {
                            const uint32_t vis = R.vis[e];
                            const uint32_t bc = occ_tab[vis >> 6];
                            o[1 + e] = (bc >> 7) * stride3 + (vis & 63u) * POINT_BYTES;      // nearest first: the n neighbours, then the pool's spare members
                        }
                    }
                }
            }
            CTGN_TICK(5)
        }
        if (!(kp.guess2 > 0.f)) break;                // no guesses: nothing to search again
        }
        after_tile(tile);
    }
    if (STAGE && !PROF && prof && lane == 0) {      // how often the prototype's path ran (ctgn_phase_cycles slots 7 .. 9)
        atomicAdd(&prof[7], (unsigned long long) g3_fills);
        atomicAdd(&prof[8], (unsigned long long) g3_rounds);
        atomicAdd(&prof[9], (unsigned long long) g3_all_rounds);
        atomicAdd(&prof[4], (unsigned long long) g3_eligible);
        atomicAdd(&prof[5], (unsigned long long) g3_points);
        atomicAdd(&prof[6], (unsigned long long) g3_over);
    }
    if (PROF && lane == 0) {
        unsigned long long tot_ = 0;
        for (int q = 0; q < 10; ++q) { atomicAdd(&prof[q], pc[q]); if (q < 7) tot_ += pc[q]; }
        atomicMax(&prof[10], tot_);            // slowest wave
        atomicAdd(&prof[11], 1ull);            // waves
        atomicAdd(&prof[12], pc[10]);          // hash probes issued (voxels not culled; the shared-home path probes once per wave)
        atomicAdd(&prof[13], pc[11]);          // map points streamed (16-point chunk slots that held a point)
        // per-wave timeline of the LAST launch: start and end clock, fast-path rounds, rounds (prof + 16, 4 per wave)
        unsigned long long *wrec = prof + 16 + 4 * (size_t) (blockIdx.x * ROW_WAVES + wave);
        wrec[0] = t_wave_start; wrec[1] = __builtin_readcyclecounter(); wrec[2] = pc[8]; wrec[3] = pc[9];
    }
#undef CTGN_TICK
}

// NB = sweep half-width (1 -> 27 voxels, 2 -> 125); HIST = histogram-assisted selection; PROF / SHARED: see rows_tiles; POOLS = compile the
// pool check (phase V) in: off for the A/B instantiations and the small-frame persistent kernel, which never see a frame large enough
// to use it and would only carry its registers.
// What k_state_init writes (below), for the kernels that do it on their way: the solve's first search launch (StateInit) — the first
// search of a solve works on the uploaded world points and reads nothing of the state, so one thread of its first block writes the state
// while the others search, and the launch in front of it (4 us + a launch boundary per solve) is gone.
struct StateInit {
    const double *pose_in;      // nullptr: the state is initialised already
    double tb, te;
};
__device__ __forceinline__ void state_init_body(GnState *st, const double *pose_in, double tb, double te) {
    Quat qb = quat_normalized(Quat{pose_in[0], pose_in[1], pose_in[2], pose_in[3]});
    Quat qe = quat_normalized(Quat{pose_in[7], pose_in[8], pose_in[9], pose_in[10]});
    st->pose[0] = qb.x; st->pose[1] = qb.y; st->pose[2] = qb.z; st->pose[3] = qb.w;
    st->pose[7] = qe.x; st->pose[8] = qe.y; st->pose[9] = qe.z; st->pose[10] = qe.w;
    for (int c = 0; c < 3; ++c) { st->pose[4 + c] = pose_in[4 + c]; st->pose[11 + c] = pose_in[11 + c]; }
    st->tbe[0] = tb; st->tbe[1] = te;
    SlerpPair sp = slerp_prepare(qb, qe);
    st->slerp_theta = sp.theta; st->slerp_sin = sp.sin_theta; st->slerp_linear = sp.linear; st->slerp_negate = sp.negate;
    for (int i = 0; i < 12; ++i) st->x[i] = 0.0;
    st->step_norm = 0.0;
    st->iter = 0; st->done = 0; st->failed = 0; st->n_used = 0;
    st->clk_iter_start = 0ull; st->ticks_neighborhood = 0ull; st->ticks_solve = 0ull; st->ticks_iter = 0ull;
}

// ABL = the ablation mask is honoured (measurement launches); false = the mask is compiled out as 0 (what a solve without a mask runs: the
// two dozen `ablate & bit` tests of the hot loops and their scalar registers are gone from the instruction stream).
template <int NB, bool HIST, bool PROF = false, int WPS = 4, bool SHARED = false, bool POOLS = true, bool STAGE = false, bool ABL = true>
__global__ __launch_bounds__(ROW_BLOCK, WPS) void k_accumulate_rows(MapView map, KpView kp, const GnState *st, GnParams prm,
                                                               double *partials, DebugView dbg, int first_iter, int rounds,
                                                               unsigned long long *prof = nullptr, int ablate_arg = 0,
                                                               StateInit init = StateInit{nullptr, 0.0, 0.0}) {
    const int ablate = ABL ? ablate_arg : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (init.pose_in) {                                // the solve's first launch: the state is written here (and not read by this launch)
        if (blockIdx.x == 0 && threadIdx.x == 0) state_init_body(const_cast<GnState *>(st), init.pose_in, init.tb, init.te);
    } else if (st->done && !(ablate & 0xffff)) return;       // an ablated run starves the solve: keep timing the search anyway
    if (blockIdx.x == 0 && threadIdx.x == 0 && kp.clk_iter_start) *kp.clk_iter_start = wall_clock64();
    const int wave = threadIdx.x >> 6;
    if (kp.n_dev) {
        // the positions to search were counted on the device (k_pool_check's fail list): tile shape from that count — every resident wave
        // one tile, as few rounds as that allows
        kp.n = min(*kp.n_dev, kp.n);
        if (kp.n <= 0) return;
        const int waves = (int) gridDim.x * ROW_WAVES;
        rounds = max(1, min(16, (kp.n + 4 * waves - 1) / (4 * waves)));
    }
    const int ntiles = (kp.n + 4 * rounds - 1) / (4 * rounds);
    // Tile hand-out. Default: wave w of block b starts at tile 4 b + w and strides by the grid. With kp.xcd_split (sorted
    // positions over a map larger than the caches) the tiles are cut into eight contiguous ranges and the blocks that land on
    // XCD x (workgroups are dealt round-robin to the 8 XCDs) stay inside range x, so each XCD's L2 holds an eighth of the
    // map region in flight instead of all eight holding the same lines.
    int tile_first = blockIdx.x * ROW_WAVES + wave, tile_end = ntiles, tile_step = gridDim.x * ROW_WAVES;
    if (kp.xcd_split) {
        const int x = blockIdx.x & 7, per = (ntiles + 7) >> 3;
        const int blocks_on_x = ((int) gridDim.x - x + 7) >> 3;
        tile_first = x * per + (blockIdx.x >> 3) * ROW_WAVES + wave;
        tile_end = min(ntiles, (x + 1) * per);
        tile_step = blocks_on_x * ROW_WAVES;
    }
    rows_tiles<NB, HIST, PROF, SHARED, POOLS, STAGE>(map, kp, st, prm, dbg, first_iter, rounds, prof, ablate, smem, tile_first, tile_end, tile_step, [](int) {});
}

// ================================================================================================
// k_pool_check — phases A and V of the row search as a kernel of their own (round 4), for the launches in which (nearly) every
// keypoint has a pool: from the third search of a solve on. What that buys:
//   * occupancy. The pool check is a chain of latencies (record -> pool members -> distances -> rank -> rewrite) with ~250 vector
//     instructions per round of four keypoints; inside k_accumulate_rows it runs at that kernel's 3 waves per SIMD (165 registers,
//     10 KB of LDS per wave for the candidate lists and probe tables it does not use). On its own it needs a third of the registers and
//     a quarter of the LDS, so twice as many waves hide each other's waits.
//   * dense search rounds. A keypoint the certificate does not cover used to be searched by its own wave right away, in rounds of four
//     filled from that wave's 4 x rounds keypoints only: at 5 % failures nearly every wave ran one round with two of its four rows idle,
//     and the launch waited for the waves that ran two or three. Here the failing positions of all waves go to ONE list (a wave-aggregated
//     atomic append; the order is irrelevant: every keypoint's result is independent, and the sums are taken later in a fixed order)
//     and k_accumulate_rows then runs over that list with every row busy and the rounds spread over the whole chip (KpView::n_dev,
//     resume): same probes, same stream, same selection, same record per keypoint.
// The certificate and its arithmetic are phase V's (rows_tiles), statement for statement; a failing keypoint leaves with the bound its
// search starts from in kth[1] (a DISTANCE: its (k + POOL_REFILL)-th pool member's, else the previous k-th distance + the distance moved)
// and kth[0] = 0 (no pool to check again).
// ================================================================================================
constexpr int CHECK_WPS = 5;                 // waves per SIMD the pool-check kernel is compiled for (<= 96 registers; at 6 it spills)
struct CheckScratch {
    double px[64], py[64], pz[64];     // world point of the tile's keypoints
    int id[64];                        // position in the working arrays; -1 = none
    float kb[64], rr2[64];             // as WaveScratch
    uint8_t todo[64];
    PoolList list[4];
    uint32_t T[4][32];                 // the row's pool: point byte offsets by pool index
};

template <bool HIST, int WPS>
__global__ __launch_bounds__(ROW_BLOCK, WPS) void k_pool_check(MapView map, KpView kp, const GnState *st, GnParams prm, int rounds, int nb_sweep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (kp.clk_iter_start) *kp.clk_iter_start = wall_clock64();
        *kp.fail_count_next = 0;                       // last touched by the search kernel of the previous iteration: stream order
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4, sub = lane & 15;
    CheckScratch &W = reinterpret_cast<CheckScratch *>(smem)[wave];
    PoolList &R = W.list[row];
    uint32_t *T = W.T[row];
    const int k = prm.max_nb;
    const char *pbase = reinterpret_cast<const char *>(map.blocks);
    const int kp_per_wave = 4 * rounds;
    const int ntiles = (kp.n + kp_per_wave - 1) / kp_per_wave;
    for (int tile = blockIdx.x * ROW_WAVES + wave; tile < ntiles; tile += gridDim.x * ROW_WAVES) {
        // ---------------- phase A (rows_tiles): lane (row, sub < rounds) owns position (sub * ntiles + tile) * 4 + row
        int my_kp = -1;
        if (sub < rounds) {
            const int pos = sub * ntiles * 4 + tile * 4 + row;
            if (pos < kp.n) my_kp = pos;
        }
        const bool own = my_kp >= 0;
        {
            Vec3 p{0, 0, 0};
            float kbv = __int_as_float(0x7f800000), rr2v = 0.f;
            if (own) {
                const Vec3 raw{kp.rx[my_kp], kp.ry[my_kp], kp.rz[my_kp]};
                const double alpha = alpha_timestamp(kp.t[my_kp], st->tbe[0], st->tbe[1]);
                const Vec3 before{kp.wx[my_kp], kp.wy[my_kp], kp.wz[my_kp]};
                p = ct_transform(st, alpha, raw);
                kp.wx[my_kp] = p.x; kp.wy[my_kp] = p.y; kp.wz[my_kp] = p.z;
                const double dx = p.x - before.x, dy = p.y - before.y, dz = p.z - before.z;
                const double moved = sqrt(sq_norm3(dx, dy, dz));
                const double rprev = (double) kp.kth[2 * my_kp], kprev = (double) kp.kth[2 * my_kp + 1];
                if (kprev > 0.0) {
                    const double reach = kprev + moved;
                    kbv = __double2float_ru(reach * reach * (1.0 + 1e-9));
                }
                const double rnow = rprev - moved * (1.0 + 1e-9) - 1e-12;
                if (rnow > 0.0) rr2v = __double2float_rd(rnow * rnow * (1.0 - 1e-9));
                const int a = voxel_coord(p.x, map.resolution), b = voxel_coord(p.y, map.resolution), c = voxel_coord(p.z, map.resolution);
                if (!(sweep_in_short_range(a, nb_sweep) && sweep_in_short_range(b, nb_sweep) && sweep_in_short_range(c, nb_sweep))) rr2v = 0.f;
            }
            W.px[lane] = p.x; W.py[lane] = p.y; W.pz[lane] = p.z;
            W.id[lane] = my_kp;
            W.kb[lane] = kbv;
            W.rr2[lane] = rr2v;
            W.todo[lane] = own ? 1 : 0;
        }
        // ---------------- phase V (rows_tiles)
        if (any64(W.rr2[lane] > 0.f)) {
            struct PoolRec { uint32_t hdr, o0, o1; };
            struct PoolPts { double x0, y0, z0, x1, y1, z1; };
            auto request = [&](int r, PoolRec &q) {
                const int id = W.id[row * 16 + min(r, rounds - 1)];
                const uint32_t *o = kp.sel + (size_t) max(id, 0) * SEL_STRIDE;
                q.hdr = o[0]; q.o0 = o[1 + sub]; q.o1 = o[17 + sub];
            };
            auto pool_size = [&](int r, const PoolRec &q) {
                return (r < rounds && W.rr2[row * 16 + min(r, rounds - 1)] > 0.f) ? (int) ((q.hdr >> 8) & REC_N_MASK) : 0;
            };
            auto gather = [&](int m, const PoolRec &q, PoolPts &t) {
                load_point(pbase, sub < m ? q.o0 : 0u, t.x0, t.y0, t.z0);
                load_point(pbase, sub + 16 < m ? q.o1 : 0u, t.x1, t.y1, t.z1);
            };
            PoolRec rec_cur{0u, 0u, 0u}, rec_nxt{0u, 0u, 0u};
            PoolPts pts_cur{0, 0, 0, 0, 0, 0};
            request(0, rec_cur);
            request(1, rec_nxt);
            gather(pool_size(0, rec_cur), rec_cur, pts_cur);
            for (int r = 0; r < rounds; ++r) {
                const int src = row * 16 + r;
                const float rr2 = W.rr2[src];
                const bool vrow = rr2 > 0.f;                                  // row-uniform
                const bool work = any64(vrow);
                const int m = pool_size(r, rec_cur);
                const bool v0 = sub < m, v1 = sub + 16 < m;
                if (work) {
                    const double qx = W.px[src], qy = W.py[src], qz = W.pz[src];
                    R.d2[sub] = sq_norm3(pts_cur.x0 - qx, pts_cur.y0 - qy, pts_cur.z0 - qz); R.vis[sub] = (uint32_t) sub;
                    R.d2[sub + 16] = sq_norm3(pts_cur.x1 - qx, pts_cur.y1 - qy, pts_cur.z1 - qz); R.vis[sub + 16] = (uint32_t) (sub + 16);
                    T[sub] = rec_cur.o0; T[sub + 16] = rec_cur.o1;
                }
                // round r's points and record words are in LDS now: their registers take the next round's points (in flight while this
                // round is ranked) and the record after that — one set of point registers instead of two
                PoolRec rec_far;
                request(r + 2, rec_far);
                gather(pool_size(r + 1, rec_nxt), rec_nxt, pts_cur);
                if (work) {
                    bool tie = false;
                    // A pool whose members are still in order — each one farther than the one before it by more than the near-tie margin —
                    // needs no ranking: the selection would return the identity and no tie flag (its float keys either differ, or the
                    // exact rank orders strictly increasing distances as they stand). From the fourth search of a solve on that is most
                    // pools of a converging scan (config D: 77 % / 96 % of the keypoints in iterations 4 / 5), and this kernel is bound by
                    // VALU issue (0.93 at five waves per SIMD), half of it the rank. Wave-uniform: all four rows or none.
                    const double a0 = R.d2[sub], a1 = R.d2[sub + 16], b0 = R.d2[sub + 1], b1 = R.d2[sub + 17];
                    const bool in_order = (sub + 1 >= m || b0 - a0 > b0 * NEAR_TIE_REL) && (sub + 17 >= m || b1 - a1 > b1 * NEAR_TIE_REL);
                    if (any64(vrow && !in_order)) row_select<HIST>(R, m, KMAX, sub, row, map.r2thr, tie);
                    const double s0 = R.d2[sub], s1 = R.d2[sub + 16];
                    const int n_in = row_sum_i32(((v0 && s0 <= map.r2thr) ? 1 : 0) + ((v1 && s1 <= map.r2thr) ? 1 : 0));   // map.h:491-493
                    const int n = min(n_in, k);
                    const double need2 = n_in >= k ? R.d2[k - 1] : map.r2thr;
                    const bool pass = vrow && need2 * (1.0 + 1e-8) < (double) rr2;
                    const int kp_r = W.id[src];
                    if (pass) {
                        uint32_t *o = kp.sel + (size_t) kp_r * SEL_STRIDE;
                        if (sub == 0) {
                            const uint32_t flagged = (uint32_t) n | ((uint32_t) m << 8) | (tie ? TIE_FLAG : 0u);
                            if (flagged != rec_cur.hdr) o[0] = flagged;
                            kp.cnt[kp_r] = flagged;
                            kp.kth[2 * kp_r] = __int_as_float(max(__float_as_int(sqrtf(rr2)) - 2, 0));          // rounded down
                            kp.kth[2 * kp_r + 1] = sqrt_bound_up(need2);
                            W.todo[src] = 0;
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int e = sub + 16 * h;
                            // nearest first: the n neighbours, then the rest of the pool — only the entries that moved
                            if (e < m && R.vis[e] != (uint32_t) e) o[1 + e] = T[R.vis[e]];
                        }
                    } else if (vrow && sub == 0 && n_in >= k + POOL_REFILL) {
                        W.kb[src] = __double2float_ru(R.d2[k + POOL_REFILL - 1] * (1.0 + 0x1p-40));
                    }
                }
                rec_cur = rec_nxt; rec_nxt = rec_far;
            }
        }
        // ---------------- the keypoints still to be searched: one append per wave
        {
            const bool fail = own && W.todo[lane] != 0;
            const unsigned long long fm = ballot64(fail);
            if (fm) {
                int base = 0;
                if (lane == 0) base = atomicAdd(kp.fail_count, (int) __popcll(fm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (fail) {
                    kp.fail_list[base + (int) __popcll(fm & ((1ull << lane) - 1ull))] = (uint32_t) my_kp;
                    const float kbv = W.kb[lane];
                    // the search's bound as a distance, rounded up (+inf: the radius only)
                    kp.kth[2 * my_kp] = 0.f;
                    kp.kth[2 * my_kp + 1] = kbv < __int_as_float(0x7f800000) ? __double2float_ru(sqrt((double) kbv) * (1.0 + 1e-12)) : 0.f;
                }
            }
        }
    }
}

template <int NB>
__host__ __device__ constexpr size_t rows_kernel_smem() {
    constexpr int S = 2 * NB + 1, V = S * S * S, OCC = (V + 3) & ~3;
    return sizeof(WaveScratch<OCC>) * ROW_WAVES;
}

// ================================================================================================
// k_residual_reduce — second half of the accumulate step for the row kernel: one lane per keypoint.
//   reads the keypoint's neighbour set (count + block-storage offsets, nearest first) left by k_accumulate_rows,
//   gathers the points, sums mean / covariance in the reference's order, 3x3 eigen-solve -> normal + a2D, gates,
//   residual, 12-vector u (ct_icp.cpp:769-841), then the packed u u^T | -u r | count per block (:843-850) as the 13 x 13
//   product U^T U of the wave's 64 records on the FP64 matrix pipe (v_mfma_f64_16x16x4_f64).
// Splitting it off keeps the neighbour-search kernel free of the eigen-solver's registers and runs this part with
// all 64 lanes busy instead of the 4 x rounds owner lanes of a tile.
// ================================================================================================
constexpr int RES_BLOCK = 256;
constexpr int GG = 10;               // neighbours gathered per group: 2 x GG independent loads in flight per lane; k = 20 (every shipped profile) is two groups
constexpr int NGROUPS = (KMAX + GG - 1) / GG;

typedef double d4_t __attribute__((ext_vector_type(4)));

// One wave, one keypoint per lane (position my_pos): neighbour set -> sums -> normal, gates, residual, u -> the wave's 13 x 13 product
// U^T U added to accm (FP64 MFMA, fixed order). `rec` is this wave's 64 x 13 LDS staging area.
// The dependent chain is what a 132 k-keypoint scan waits for (2 068 waves on 1 024 SIMDs: two per SIMD, little to overlap with), so
// everything that does not depend on something else is requested up front: the dense count, the record's first 1 + k words (k is
// wave-uniform: six 16-byte loads at k = 20, not nine) and the keypoint's own seven doubles go out together; then the gathers, GG at a time.
__device__ __forceinline__ void residual_tile(const MapView &map, const KpView &kp, const GnState *st, const GnParams &prm, const DebugView &dbg,
                                              int ablate, int my_pos, int lane, double *rec, d4_t &accm, int &n_used_wave, TieScratch &tie) {
    const char *pbase = reinterpret_cast<const char *>(map.blocks);
        // position -> keypoint: with kp.order the lanes of a wave take keypoints of neighbouring voxels, so their gathers share
        // cache lines (the sums then run in position order: still fixed, a different rounding than index order)
        const int my_kp = (kp.order && my_pos < kp.n) ? (int) kp.order[my_pos] : my_pos;
        double u[12], rr = 0.0;
        bool used = false;
        if (my_kp < kp.n) {
            int res_n;
            bool fetch_rec;
            Vec3 res_S{0, 0, 0}, res_q{0, 0, 0};
            Sym3 res_SS{0, 0, 0, 0, 0, 0};
            const uint32_t cnt_raw = kp.cnt[my_kp];
            uint32_t rec32[SEL_STRIDE];
            {
                const uint4 *in4 = reinterpret_cast<const uint4 *>(kp.sel + (size_t) my_kp * SEL_STRIDE);
                const int nq = dbg.n_nb != nullptr ? SEL_STRIDE / 4 : min(SEL_STRIDE / 4, (prm.max_nb + 4) >> 2);      // words 0 .. k
#pragma unroll
                for (int q = 0; q < SEL_STRIDE / 4; ++q) {
                    uint4 v4 = make_uint4(0u, 0u, 0u, 0u);
                    if (q < nq) v4 = in4[q];
                    rec32[4 * q] = v4.x; rec32[4 * q + 1] = v4.y; rec32[4 * q + 2] = v4.z; rec32[4 * q + 3] = v4.w;
                }
            }
            const Vec3 raw{kp.rx[my_kp], kp.ry[my_kp], kp.rz[my_kp]};
            const Vec3 p{kp.wx[my_kp], kp.wy[my_kp], kp.wz[my_kp]};       // written (or taken as given) by phase A of the search kernel
            const double t_kp = kp.t[my_kp];
            // a keypoint the gates drop on its count alone (fewer than min_number_neighbors, or 5: ct_icp.cpp:769, neighborhood.h:227) costs
            // nothing further, unless debug capture wants its farthest neighbour
            const int cnt_n = min((int) (cnt_raw & REC_N_MASK), KMAX);
            fetch_rec = (cnt_n >= prm.min_nb && cnt_n >= 5) || dbg.n_nb != nullptr;
            rec32[0] = (uint32_t) cnt_n | (cnt_raw & TIE_FLAG);
            resolve_ties(map, kp, my_kp, fetch_rec, rec32, tie, lane, prm.max_nb);      // rare: see TIE_FLAG
            res_n = (ablate & 4) ? 0 : min((int) rec32[0], KMAX);
            const int gat_n = ((res_n >= prm.min_nb && res_n >= 5) || dbg.n_nb != nullptr) ? res_n : 0;
            // mean / covariance sums over the kept neighbours in the reference's order: its neighbour vector is
            // farthest-first (map.h:508-513) and ComputeNeighborhood sums it front to back (neighborhood.h:236-240) — the
            // record is nearest-first, so it is walked from entry n - 1 down to entry 0. Sums strictly in order, products and sums
            // rounded separately (no fused multiply-add: the reference build has none, and these sums feed the gates).
#pragma unroll
            for (int g = NGROUPS - 1; g >= 0; --g) {
                if (GG * g < gat_n) {
                    double gx[GG], gy[GG], gz[GG];
#pragma unroll
                    for (int q = 0; q < GG; ++q) {
                        if (GG * g + q < KMAX) {
                            const uint32_t off = (GG * g + q < gat_n) ? rec32[1 + GG * g + q] : 0u;
                            load_point(pbase, off, gx[q], gy[q], gz[q]);          // one point = 24 contiguous bytes: two loads, one or two lines
                        }
                    }
#pragma unroll
                    for (int q = GG - 1; q >= 0; --q) {
                        if (GG * g + q < KMAX && GG * g + q < gat_n) {
#pragma clang fp contract(off)
                            const double x = gx[q], y = gy[q], z = gz[q];
                            if (GG * g + q == gat_n - 1) res_q = Vec3{x, y, z};      // points[0]: the farthest kept (ct_icp.cpp:791)
                            res_S.x += x; res_S.y += y; res_S.z += z;
                            const double xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
                            res_SS.xx += xx; res_SS.xy += xy; res_SS.xz += xz;
                            res_SS.yy += yy; res_SS.yz += yz; res_SS.zz += zz;
                        }
                    }
                }
            }
            Vec3 nrm{0, 0, 0};
            double a2d = 0.0;
            if (fetch_rec && !(ablate & 8)) {
                const double alpha = alpha_timestamp(t_kp, st->tbe[0], st->tbe[1]);
                used = residual_jacobian(res_n, res_S, res_SS, res_q, p, raw, alpha, st, prm, normals_mode(ablate, prm.normals), u, rr, nrm, a2d);
            }
            if (dbg.n_nb) {
                dbg.n_nb[my_kp] = res_n;
                dbg.normal[3 * my_kp] = nrm.x; dbg.normal[3 * my_kp + 1] = nrm.y; dbg.normal[3 * my_kp + 2] = nrm.z;
                dbg.a2d[my_kp] = a2d;
                dbg.farthest[3 * my_kp] = res_q.x; dbg.farthest[3 * my_kp + 1] = res_q.y; dbg.farthest[3 * my_kp + 2] = res_q.z;
                dbg.used[my_kp] = used ? 1 : 0;
            }
        }
        // packed u u^T | -u r | count. With U the wave's 64 x 13 matrix of records (u | r), the sums are the 13 x 13 product
        // U^T U: sixteen v_mfma_f64_16x16x4_f64 per tile, K = 4 keypoints each; lane l feeds U[k0 + (l >> 4)][l & 15] as both
        // A[i = l & 15][k] and B[k][j = l & 15] (one LDS read per lane and step — the lane-per-entry loop this replaces read
        // four doubles per lane and keypoint and was bound by LDS bandwidth). Accumulation order is fixed: deterministic.
        double *my = rec + lane * 13;
        const unsigned long long used_lanes = ballot64(used);
        n_used_wave += __popcll(used_lanes);
        if (used_lanes == 0ull) return;              // a wave of dropped keypoints adds exact zeros: nothing to stage, nothing to multiply
#pragma unroll
        for (int c = 0; c < 12; ++c) my[c] = used ? u[c] : 0.0;
        my[12] = used ? rr : 0.0;
        if (!(ablate & 128)) {
            const int comp = lane & 15;
#pragma unroll 4
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const double v = (comp < 13) ? rec[(k0 + (lane >> 4)) * 13 + comp] : 0.0;
                accm = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, accm, 0, 0, 0);
            }
        }
}

// lane l holds (U^T U)[row = (l >> 4) + 4 m][col = l & 15], m = 0..3 (the f64 MFMA's C/D layout); the upper triangle goes to the
// packed entries, column 12 (sum u_i r) to -J^T r, the wave's keypoint count to entry 90
__device__ __forceinline__ void unpack_wave_sums(int lane, const d4_t &accm, int n_used_wave, double *comb) {
    if (lane < SYS_N - SYS_USED) comb[SYS_USED + lane] = 0.0;
    const int col = lane & 15;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = (lane >> 4) + 4 * m;
        if (row < 12 && col >= row && col < 12) comb[row * 12 - (row * (row - 1)) / 2 + (col - row)] = accm[m];
        if (row < 12 && col == 12) comb[78 + row] = -accm[m];
    }
    if (lane == 0) comb[90] = (double) n_used_wave;
}

// loads / stores that bypass this CU's L1 (sc1: served by the XCD's L2, the point of coherence of its 32 CUs)
__device__ __forceinline__ void sc1_store(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long) __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double sc1_load(const double *p) {
    return __longlong_as_double((long long) __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Per-XCD pre-sums of the per-block records (round 5). The solve kernel's reduction pulls every block record — 518 x 768 B on a
// 132 k-keypoint sweep — through ONE compute unit: ~4.5 us of its 13.7. The workgroups of a launch are dealt round-robin to the eight
// XCDs (workgroup b runs on XCD b % 8), and the blocks of one XCD share its L2. The blocks form XCD_GROUPS = 32 groups (b % 32: four per
// XCD): each block stores its record plainly (write-through L1, the line stays in that L2), drains its stores and takes a ticket on its
// group's counter; the group's LAST block sums the group's records in block order (sc1 loads: its own L1 bypassed, L2 hits; one trip,
// every load in flight) into the group's record — 32 blocks doing that side by side behind the last of their group, each out of its own
// L2, instead of one block behind the launch boundary. The solve kernel then adds 32 records. The idiom is k_gn_persistent's, and so is
// the placement check, which must not depend on what a possibly foreign L2 returns: the ticket is ONE 64-bit atomic add (device-coherent,
// like every atomic here) whose low byte counts arrivals and whose eight 7-bit fields above count them per XCC the block ran on
// (HW_REG_XCC_ID); the last arriver sees every member's XCC in the value it gets back, and a group that arrived from more than one XCC
// stamps the launch's epoch into the control block instead of its own stamp. The solve kernel takes the group records only when every
// group stamped this launch's epoch and none flagged its placement — otherwise the per-block records (always written, as before).
// Whatever the dispatcher does, the result is one of the two fixed-order sums, never a partial or a stale one.
constexpr int XCD_GROUPS = 32;
struct XcdReduce {
    unsigned int *ctl;       // group g owns the 128-byte line ctl[32 g ..]: words 0-1 = the 64-bit ticket (arrivals | per-XCC arrivals), word 2 = epoch
                             // its record was last written in; ctl[32 XCD_GROUPS] = epoch of the last launch whose placement check failed,
                             // [+1] = 1 iff the last solve launch summed the group records
    double *rec;             // [XCD_GROUPS][SYS_N] group records
    unsigned int epoch;      // this launch's number (never 0)
};
constexpr int XCD_CTL_WORDS = 32 * XCD_GROUPS + 4;
static_assert(MAX_PARTIAL_BLOCKS / XCD_GROUPS <= 127, "a group's arrivals must fit the ticket's 7-bit per-XCC fields (and its 8-bit count)");

// BLK = 256 (throughput: 4 waves share a CU's texture path) or 64 (small frames: the 60 scattered gathers per keypoint are bound by the
// per-CU texture path — ~1 line per clock — so a 1 k-keypoint frame is spread over 16 CUs instead of 4)
template <int BLK>
__global__ __launch_bounds__(BLK, 3) void k_residual_reduce(MapView map, KpView kp, const GnState *st, GnParams prm,
                                                            double *partials, DebugView dbg, int ablate, XcdReduce xr) {
    __shared__ double s_rec[BLK / 64][64 * 13];
    __shared__ double s_comb[BLK / 64][SYS_N];
    __shared__ TieScratch s_tie[BLK / 64];
    if (st->done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    d4_t accm = {0.0, 0.0, 0.0, 0.0};
    int n_used_wave = 0;
    const int ntiles = (kp.n + BLK - 1) / BLK;
    // Which block takes which tile only decides where the gathers are issued from (and the fixed order of the partial sums). With
    // kp.xcd_split (positions sorted by home voxel over a map larger than the caches) the blocks that land on XCD x (workgroups are
    // dealt round-robin to the 8 XCDs) stay inside the x-th eighth of the positions, so each L2 fetches an eighth of the map region
    // once instead of all eight fetching all of it: workload D, 3.3 GB of HBM reads per launch for a 0.4 GB map otherwise.
    int tile = blockIdx.x, tile_end = ntiles, tile_step = gridDim.x;
    if (kp.xcd_split && gridDim.x >= 8) {
        const int x = blockIdx.x & 7, per = (ntiles + 7) >> 3;
        tile = x * per + (int) (blockIdx.x >> 3);
        tile_end = min(ntiles, (x + 1) * per);
        tile_step = ((int) gridDim.x - x + 7) >> 3;
    }
    for (; tile < tile_end; tile += tile_step)
        residual_tile(map, kp, st, prm, dbg, ablate, tile * BLK + tid, lane, s_rec[wave], accm, n_used_wave, s_tie[wave]);
    unpack_wave_sums(lane, accm, n_used_wave, s_comb[wave]);
    __syncthreads();
    for (int e = tid; e < SYS_N; e += BLK) {
        double s = 0.0;
        for (int w = 0; w < BLK / 64; ++w) s += s_comb[w][e];
        partials[(size_t) blockIdx.x * SYS_N + e] = s;          // block-major: one contiguous 768-byte record per block (see reduce_partials)
    }
    if constexpr (BLK >= 2 * SYS_N) {
        if (xr.ctl) {            // per-XCD pre-sum (XcdReduce)
            __shared__ int s_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's part of the record has reached L2
            __syncthreads();
            const int g = (int) (blockIdx.x % XCD_GROUPS);
            const int members = ((int) gridDim.x - g + XCD_GROUPS - 1) / XCD_GROUPS;
            if (tid == 0) {
                // one 64-bit add: arrivals in bits 0-7 (a group has at most 64 members), arrivals from XCC x in bits 8 + 7 x .. 14 + 7 x
                const unsigned int my_xcc = (unsigned int) __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;       // HW_REG_XCC_ID
                unsigned long long *ticket = reinterpret_cast<unsigned long long *>(xr.ctl + 32 * g);
                const unsigned long long inc = 1ull + (1ull << (8 + 7 * my_xcc));
                const unsigned long long t = __hip_atomic_fetch_add(ticket, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int last = 0;
                if ((t & 0xffull) == (unsigned long long) (members - 1)) {
                    __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                  // for the next launch (stream order)
                    const unsigned long long by_xcc = (t + inc) >> 8;
                    if ((by_xcc & ~(0x7full << (7 * my_xcc))) != 0ull)                                              // members behind another L2: void
                        __hip_atomic_store(xr.ctl + 32 * XCD_GROUPS, xr.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else last = 1;
                }
                s_last = last;
            }
            __syncthreads();
            if (s_last) {
                // thread (h, e), h = 0 / 1: entry e of the group's members h, h + 2, ... in that order, every load in flight at once (a group
                // of a resident grid has at most 24 members: 12 per thread); then half 0 + half 1
                constexpr int MAXM = (3 * 256 + XCD_GROUPS - 1) / XCD_GROUPS;      // members of a group at the residual kernel's largest grid (3 blocks per CU)
                if (tid < 2 * SYS_N) {
                    const int hh = tid / SYS_N, e2 = tid - hh * SYS_N;
                    const double *col = partials + (size_t) g * SYS_N + e2;          // member m = block g + XCD_GROUPS m
                    double sum = 0.0;
                    for (int m0 = hh; m0 < members; m0 += MAXM) {
                        double v[(MAXM + 1) / 2];
#pragma unroll
                        for (int q = 0; q < (MAXM + 1) / 2; ++q) v[q] = (m0 + 2 * q < members) ? sc1_load(col + (size_t) (m0 + 2 * q) * XCD_GROUPS * SYS_N) : 0.0;
#pragma unroll
                        for (int q = 0; q < (MAXM + 1) / 2; ++q) sum += v[q];
                    }
                    s_comb[hh][e2] = sum;
                }
                __syncthreads();
                if (tid < SYS_N) xr.rec[(size_t) g * SYS_N + tid] = s_comb[0][tid] + s_comb[1][tid];
                if (tid == 0) __hip_atomic_store(xr.ctl + 32 * g + 2, xr.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // this group's record is this launch's
            }
        }
    }
}

// ================================================================================================
// k_search_residual — small frames (round 4): the search of a tile and, in the same wave right behind it, the residual part for the same
// keypoints (rows_tiles' after_tile hook, as in k_gn_persistent, without that kernel's in-kernel barrier and solve): an iteration is two
// launches instead of three, and a keypoint's neighbour record never leaves the CU that wrote it. For frames the three-launch loop
// spreads thinly anyway (a few waves per CU: nothing for a separate, fuller residual kernel to gain); 255 registers, two waves per SIMD.
// Same per-keypoint arithmetic; the packed sums are taken per wave, per block, blocks in index order (a fixed order of its own).
// ================================================================================================
struct SearchResidualShared {
    double comb[ROW_WAVES][SYS_N];
    TieScratch tie[ROW_WAVES];
};
template <int NB>
inline size_t search_residual_smem() { return rows_kernel_smem<NB>() + sizeof(SearchResidualShared); }

template <int NB>
__global__ __launch_bounds__(ROW_BLOCK, 2) void k_search_residual(MapView map, KpView kp, const GnState *st, GnParams prm, double *partials,
                                                                  DebugView dbg, int first_iter, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && kp.clk_iter_start) *kp.clk_iter_start = wall_clock64();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SearchResidualShared &P = *reinterpret_cast<SearchResidualShared *>(smem + rows_kernel_smem<NB>());
    constexpr int OCC = ((2 * NB + 1) * (2 * NB + 1) * (2 * NB + 1) + 3) & ~3;
    WaveScratch<OCC> &W = reinterpret_cast<WaveScratch<OCC> *>(smem)[wave];
    const int ntiles = (kp.n + 4 * rounds - 1) / (4 * rounds);
    d4_t accm = {0.0, 0.0, 0.0, 0.0};
    int n_used_wave = 0;
    rows_tiles<NB, true, false, false, false, false>(map, kp, st, prm, dbg, first_iter, rounds, nullptr, 0, smem, blockIdx.x * ROW_WAVES + wave, ntiles,
                                       gridDim.x * ROW_WAVES, [&](int) {
        const int id = W.id[lane];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the wave's own records and world points have left for L2
        residual_tile(map, kp, st, prm, dbg, 0, id < 0 ? kp.n : id, lane, W.rec, accm, n_used_wave, P.tie[wave]);
    });
    unpack_wave_sums(lane, accm, n_used_wave, P.comb[wave]);
    __syncthreads();
    for (int e = tid; e < SYS_N; e += ROW_BLOCK) {
        double sum = 0.0;
        for (int w = 0; w < ROW_WAVES; ++w) sum += P.comb[w][e];
        partials[(size_t) blockIdx.x * SYS_N + e] = sum;
    }
}

// ================================================================================================
// k_reduce_solve — partials -> packed system -> (normalise, motion prior, LDL^T, pose update, stop test)
//   mode 0: reduce + solve   1: reduce only (multi-GPU: the all-reduce sits in between)   2: solve only
// One block of 1024 threads. Reduction: wave w sums entries w, w+16, ... over the blocks with a fixed
// lane-strided order and a fixed shuffle tree (deterministic). Solve: the 12 x 12 system lives in LDS and lanes
// 0..11 of wave 0 own one row each (Eigen's diagonally pivoted, left-looking LDL^T, ct_icp.cpp:914).
// ================================================================================================
constexpr int SOLVE_BLOCK = 1024;

__device__ __forceinline__ double wave_sum_fixed(double v) {
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);  v += __shfl_xor(v, 2);  v += __shfl_xor(v, 1);
    return v;
}

#define WSYNC() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

struct SolveScratch {                 // LDS of the 12 x 12 solve (wave 0)
    double sys[SYS_N];
    double m[144], temp[12], x[12], b[12], sc[12], q[8];
    int perm[12];
};

// Sum of the per-block partials of the packed system. Layout since round 3: BLOCK-major, partials[block][96] — a block's record is 768
// contiguous bytes, written with two coalesced wave stores and spread over the L2 channels. (Rounds 1-2 stored entry-major,
// [96][2048]: the 96 columns lie 16 KB apart, which is one and the same L2 channel set for every one of them — each block's 96
// eight-byte writes and every reader's loads queued there; found while timing the persistent kernel's exchange)
// THREADS / 96 groups of 96 threads: thread (g, e) adds entry e of blocks g, g + G, g + 2G, ... in that order (coalesced across e,
// sixteen loads in flight), the groups' sums are added in group order. Fixed order: deterministic. s_tmp: G x 96 doubles of LDS.
template <int THREADS>
__device__ __forceinline__ void reduce_partials(const double *partials, int nblocks, int tid, double *sys_global, double *s_sys, double *s_tmp) {
    constexpr int G = THREADS / SYS_N;
    static_assert(G >= 1, "at least one group of 96 threads");
    if (tid < G * SYS_N) {
        const int g = tid / SYS_N, e = tid - g * SYS_N;
        const double *col = partials + e;
        double sum = 0.0;
        // sixteen loads in flight per thread (a 132 k-keypoint scan leaves ~26 records per thread: two trips instead of four); the tail
        // batch is masked — adding 0.0 keeps the order and the value of the sum
        for (int b = g; b < nblocks; b += 16 * G) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = (b + q * G < nblocks) ? col[(size_t) (b + q * G) * SYS_N] : 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += v[q];
        }
        s_tmp[g * SYS_N + e] = sum;
    }
    __syncthreads();
    if (tid < SYS_N) {
        double sum = 0.0;
#pragma unroll
        for (int g = 0; g < G; ++g) sum += s_tmp[g * SYS_N + tid];
        if (tid >= SYS_USED) sum = 0.0;            // pad entries
        sys_global[tid] = sum;
        s_sys[tid] = sum;
    }
}

// GnState::failed: 1 = fewer than min_used keypoints (the reference's soft failure), 3 = GN_FAILED_BARRIER (persistent kernel),
// 4 = a peer rank of the keypoint-sharded mode failed before the exchange (it takes part in every all-reduce with a count of
// -1e300, so that every rank stops with the same error instead of waiting for it for ever)
constexpr int GN_FAILED_PEER = 4;
constexpr double GN_PEER_POISON = -1e300;

// normalise, motion prior, LDL^T, pose update, stop test (ct_icp.cpp:860-962, :978-980) by ONE wave on the packed system in S.sys
__device__ __forceinline__ void solve_wave0(SolveScratch &S, GnState *st, const GnParams &prm, int min_used, int lane,
                                            unsigned long long tc0, unsigned long long wall0) {
    double *s_sys = S.sys, *s_m = S.m, *s_temp = S.temp, *s_x = S.x, *s_b = S.b, *s_sc = S.sc, *s_q = S.q;
    int *s_perm = S.perm;
    const unsigned long long tc1 = __builtin_readcyclecounter();
    if (s_sys[90] < -0.5) {               // keypoint-sharded mode: a peer could not start its solve and poisoned the summed count
        if (lane == 0) { st->n_used = 0; st->failed = GN_FAILED_PEER; st->done = 1; }
        return;
    }
    const int n_used = (int) (s_sys[90] + 0.5);
    if (n_used < min_used) {              // ct_icp.cpp:860-871 — soft failure, pose untouched
        if (lane == 0) { st->n_used = n_used; st->failed = 1; st->done = 1; }
        return;
    }
    const double dn = (double) n_used;
    // A = sum / n (ct_icp.cpp:877-882), full symmetric matrix in LDS; b in a register of lane i
    for (int e = lane; e < 78; e += 64) {
        const int i = c_tri_i[e], j = c_tri_j[e];
        const double v = s_sys[e] / dn;
        s_m[12 * i + j] = v;
        s_m[12 * j + i] = v;
    }
    double bi = (lane < 12) ? s_sys[78 + lane] / dn : 0.0;
    WSYNC();
    if (prm.has_prior && lane < 3) {                              // ct_icp.cpp:885-910
        const int c = lane;
        s_m[13 * (3 + c)] += prm.beta_c;
        s_m[13 * (9 + c)] += prm.beta_e;
    }
    if (prm.has_prior) {
        if (lane >= 3 && lane < 6) {
            const int c = lane - 3;
            bi -= prm.beta_c * (st->pose[4 + c] - st->pose[11 + c]);
        } else if (lane >= 9 && lane < 12) {
            const int c = lane - 9;
            bi -= prm.beta_e * (st->pose[11 + c] - st->pose[4 + c] - prm.prev_e[c] + prm.prev_b[c]);
        }
    }
    WSYNC();
    // LDS traffic of this single wave executes in program order; WSYNC only stops the compiler from moving
    // accesses across the phase boundaries (no volatile: reads inside a phase can be batched)
    double *m = s_m;
    double *temp = s_temp;
    int *perm = s_perm;                   // perm[p] = original index of the element now at position p
    const int i = lane;                   // row owned by this lane (valid for lane < 12)
    if (i < 12) { perm[i] = i; s_b[i] = bi; }
    WSYNC();
    // Fast factorisation: un-pivoted, right-looking LDL^T with row i of the (symmetric) matrix in the registers of lane
    // i; the pivot row is broadcast with v_readlane, no LDS traffic. For a positive definite system it agrees with the
    // pivoted factorisation to rounding. Taken only when every pivot is safely positive relative to the largest
    // diagonal entry; otherwise the Eigen-style diagonally pivoted factorisation below runs on the untouched s_m.
    bool fast_ok = true;
    {
        const int ii0 = (i < 12) ? i : 0;
        double rowr[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) rowr[j] = m[12 * ii0 + j];
        double dmax = 0.0;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const double djj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rowr[j]), j),
                                                __builtin_amdgcn_readlane(__double2loint(rowr[j]), j));
            dmax = fmax(dmax, fabs(djj));
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const double dk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rowr[k]), k),
                                               __builtin_amdgcn_readlane(__double2loint(rowr[k]), k));
            fast_ok = fast_ok && (dk > 1e-10 * dmax);
            const double lik = (i > k) ? rowr[k] / dk : 0.0;
#pragma unroll
            for (int j = k + 1; j < 12; ++j) {
                const double vkj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rowr[j]), k),
                                                    __builtin_amdgcn_readlane(__double2loint(rowr[j]), k));
                if (i > k) rowr[j] -= lik * vkj;
            }
            if (i > k) rowr[k] = lik;
        }
        fast_ok = fast_ok && (dmax > 0.0);
        if (fast_ok && i < 12) {
#pragma unroll
            for (int j = 0; j < 12; ++j) m[12 * i + j] = rowr[j];       // lower triangle: L, diagonal: D (upper: scratch)
        }
    }
    WSYNC();
    for (int k = 0; k < (fast_ok ? 0 : 12); ++k) {
        // pivot: first index of the largest |diagonal| among k..11 — DPP butterfly over lanes 0..15 (no LDS round trip)
        double dv = (i >= k && i < 12) ? fabs(m[13 * i]) : -1.0;
        int di = i;
#define CTGN_PIVOT_STEP(CTRL)                                                                    \
        {                                                                                        \
            const double ov = dpp_f64<CTRL>(dv);                                                 \
            const int oi = __builtin_amdgcn_update_dpp(0, di, CTRL, 0xf, 0xf, false);            \
            if (ov > dv || (ov == dv && oi < di)) { dv = ov; di = oi; }                          \
        }
        CTGN_PIVOT_STEP(0xB1) CTGN_PIVOT_STEP(0x4E) CTGN_PIVOT_STEP(0x141) CTGN_PIVOT_STEP(0x140)
#undef CTGN_PIVOT_STEP
        const int big = __builtin_amdgcn_readfirstlane(di);
        if (big != k) {                   // symmetric swap on the lower triangle
            if (i < k) { double t = m[12 * k + i]; m[12 * k + i] = m[12 * big + i]; m[12 * big + i] = t; }
            if (i > big && i < 12) { double t = m[12 * i + k]; m[12 * i + k] = m[12 * i + big]; m[12 * i + big] = t; }
            if (i > k && i < big) { double t = m[12 * i + k]; m[12 * i + k] = m[12 * big + i]; m[12 * big + i] = t; }
            if (i == 0) {
                double t = m[13 * k]; m[13 * k] = m[13 * big]; m[13 * big] = t;
                int pk = perm[k]; perm[k] = perm[big]; perm[big] = pk;
            }
        }
        WSYNC();
        if (i < k) temp[i] = m[13 * i] * m[12 * k + i];
        WSYNC();
        if (i >= k && i < 12) {
            double a2 = 0.0;
            for (int j = 0; j < k; ++j) a2 += m[12 * i + j] * temp[j];
            m[12 * i + k] -= a2;
        }
        WSYNC();
        const double akk = m[13 * k];
        if (i > k && i < 12 && fabs(akk) > 0.0) m[12 * i + k] /= akk;
        WSYNC();
    }
    const unsigned long long tc2 = __builtin_readcyclecounter();
    // solve: y = P b ; L^-1 ; D^-1 ; L^-T ; x = P^T y. Lane i keeps y_i, row i and column i of L in registers and the
    // running y_j is broadcast with readlane (SALU) — no LDS traffic in the substitutions.
    const int ii = (i < 12) ? i : 0;
    double y = s_b[perm[ii]];
    double Lrow[12], Lcol[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { Lrow[j] = m[12 * ii + j]; Lcol[j] = m[12 * j + ii]; }
    const double dii = m[13 * ii];
#pragma unroll
    for (int j = 0; j < 12; ++j) {          // forward: after step j, y_j is final
        const double yj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), j),
                                           __builtin_amdgcn_readlane(__double2loint(y), j));
        if (i > j) y -= Lrow[j] * yj;
    }
    y = (fabs(dii) > DBL_MIN) ? y / dii : 0.0;
#pragma unroll
    for (int j = 11; j >= 0; --j) {         // backward with L^T
        const double yj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), j),
                                           __builtin_amdgcn_readlane(__double2loint(y), j));
        if (i < j) y -= Lcol[j] * yj;
    }
    if (i < 12) s_x[perm[i]] = y;
    WSYNC();
    const unsigned long long tc3 = __builtin_readcyclecounter();
    // pose update (ct_icp.cpp:916-962), spread over a few lanes: the six sin/cos pairs on lanes 0..5, the begin and
    // end quaternion updates on lanes 0 and 1, the bookkeeping on lane 0
    if (lane < 6) {
        double sn, cs;
        sincos(s_x[lane < 3 ? lane : lane + 3], &sn, &cs);
        s_sc[2 * lane] = sn;
        s_sc[2 * lane + 1] = cs;
    }
    WSYNC();
    if (lane < 2) {
        const int e = lane;               // 0: begin pose, 1: end pose
        const double *sc = s_sc + 6 * e;
        double Rd[9], Q[9], P[9];
        euler_rotation_sc(sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], Rd);          // :919-932 / :939-947
        const Quat q0{st->pose[7 * e], st->pose[7 * e + 1], st->pose[7 * e + 2], st->pose[7 * e + 3]};
        quat_to_matrix(q0, Q);
        mat3_mul(Rd, Q, P);
        const Quat q = quat_normalized(matrix_to_quat(P));                        // :950-955, :961-962
        st->pose[7 * e] = q.x; st->pose[7 * e + 1] = q.y; st->pose[7 * e + 2] = q.z; st->pose[7 * e + 3] = q.w;
        for (int c = 0; c < 3; ++c) st->pose[7 * e + 4 + c] += s_x[6 * e + 3 + c];
        s_q[4 * e] = q.x; s_q[4 * e + 1] = q.y; s_q[4 * e + 2] = q.z; s_q[4 * e + 3] = q.w;
    }
    WSYNC();
    if (lane != 0) return;
    st->n_used = n_used;
    const Quat qb{s_q[0], s_q[1], s_q[2], s_q[3]}, qe{s_q[4], s_q[5], s_q[6], s_q[7]};
    SlerpPair sp = slerp_prepare(qb, qe);
    st->slerp_theta = sp.theta; st->slerp_sin = sp.sin_theta; st->slerp_linear = sp.linear; st->slerp_negate = sp.negate;
    double nrm = 0.0;
    for (int c = 0; c < 12; ++c) { const double xc = s_x[c]; nrm += xc * xc; st->x[c] = xc; }
    nrm = sqrt(nrm);
    st->step_norm = nrm;
    st->iter += 1;
    if (nrm < prm.thr_norm) st->done = 1;                           // :978-980
    st->solve_cycles[0] = tc1 - tc0; st->solve_cycles[1] = tc2 - tc1; st->solve_cycles[2] = tc3 - tc2;
    st->solve_cycles[3] = __builtin_readcyclecounter() - tc3;
    // search + residual of this iteration ran from clk_iter_start to this kernel's start; the solve from there to now
    const unsigned long long wall1 = wall_clock64(), it0 = st->clk_iter_start;
    if (it0 != 0ull && wall0 >= it0) { st->ticks_neighborhood += wall0 - it0; st->ticks_iter += wall1 - it0; }
    st->ticks_solve += wall1 - wall0;
}

// BLKS = 1024 (16 waves, three 128-block spans in flight per pass) for the thousands of partial columns of a large scan; BLKS = 256
// (4 waves, 24 entries each, one span) when a small frame leaves at most 128 columns: a quarter of the waves to dispatch and to meet
// at the barrier before the solve can start.
// stop_flag (host memory the device can write, or nullptr) / stop_word: when the solve of this launch is done, lane 0 publishes
// stop_word | (stopped ? 2 : 1) there — the host, which enqueues iterations ahead of the device, polls it two iterations behind and
// stops enqueueing launches that would only find the stop flag set (ctgn_solve).
template <int BLKS>
__global__ __launch_bounds__(BLKS) void k_reduce_solve(const double *partials, int nblocks, double *sys, GnState *st,
                                                       GnParams prm, int mode, int min_used, XcdReduce xr, unsigned int *stop_flag = nullptr,
                                                       unsigned int stop_word = 0u) {
    __shared__ SolveScratch S;
    __shared__ double s_tmp[(BLKS / SYS_N) * SYS_N];
    if (st->done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long tc0 = __builtin_readcyclecounter();
    const unsigned long long wall0 = wall_clock64();
    if (mode != 2) {
        // the residual kernel left eight per-XCD group records (XcdReduce) unless its placement check failed in this very launch
        // (every wave looks at all the stamps at once, lane l at group l % 32's: one round trip — a chain of `&&` over 32 loads cost the
        // kernel 8 us, more than the reduction it replaces)
        bool groups = false;
        if (xr.ctl != nullptr && nblocks >= XCD_GROUPS) {
            const unsigned int stamp = xr.ctl[32 * (lane & (XCD_GROUPS - 1)) + 2], bad = xr.ctl[32 * XCD_GROUPS];
            groups = ballot64(stamp == xr.epoch && bad != xr.epoch) == ~0ull;
        }
        if (xr.ctl != nullptr && tid == 0) xr.ctl[32 * XCD_GROUPS + 1] = groups ? 1u : 0u;       // which sum this launch took (ctgn_path_counters)
        if (groups) {
            if (tid < SYS_N) {
                double v[XCD_GROUPS];
#pragma unroll
                for (int g = 0; g < XCD_GROUPS; ++g) v[g] = xr.rec[(size_t) g * SYS_N + tid];
                double sum = 0.0;
#pragma unroll
                for (int g = 0; g < XCD_GROUPS; ++g) sum += v[g];
                sys[tid] = sum;
                S.sys[tid] = sum;
            }
        } else {
            reduce_partials<BLKS>(partials, nblocks, tid, sys, S.sys, s_tmp);
        }
    } else {
        if (tid < SYS_N) S.sys[tid] = sys[tid];
    }
    __syncthreads();
    if (mode == 1 || wave != 0) return;
    solve_wave0(S, st, prm, min_used, lane, tc0, wall0);
    if (stop_flag != nullptr && lane == 0)
        __hip_atomic_store(stop_flag, stop_word | (st->done ? 2u : 1u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#undef WSYNC

// ================================================================================================
// k_gn_persistent — ONE launch per registration for small frames (the reference's own regime: 1-3 k keypoints x 5 iterations,
// config/odometry/driving_config.yaml): state init, every GN iteration (search -> residual -> sums -> 12 x 12 solve -> pose
// update -> stop test, ct_icp.cpp:745-981) and the final re-transform, instead of three dependent launches per iteration.
//
// Why it can be cheap on this part: the blocks that do the work all sit on ONE XCD (workgroup b is dealt to XCD b % 8: the grid is
// 8 x nblk blocks and the seven blocks of every eight that land elsewhere leave at once), so the one exchange per iteration — the
// per-block packed sums, 96 doubles each — is served by that XCD's L2, the point of coherence of its 32 CUs: publishers store plainly
// (write-through L1, the line STAYS in this L2), drain their stores (s_waitcnt vmcnt(0)) and arrive on a counter; readers poll the
// counter and read the sums with sc1 loads (their own L1 bypassed, L2 hit). No agent-scope release / acquire fence, hence no L2
// write-back or invalidate — what made the "last block reduces" variants of rounds 1-2 slower than a launch boundary — and no sc1
// (write-through-to-memory) stores either, which drop the line from L2 and made every reader wait for the fabric (measured: 24 us
// per iteration in barrier + reduce + solve against 9.5 us for the separate solve kernel).
// Placement is then a correctness matter, so it is CHECKED, not assumed: every working block ORs the XCC it runs on (HW_REG_XCC_ID)
// into a mask with its first arrival; more than one bit set means the dispatcher spread the blocks over several L2s, the kernel
// stops before anything is read through the wrong cache and the host falls back to the three-launch loop for good.
// Every block then reduces ALL published sums in the same fixed order and runs the same solve on its own copy of the state (LDS):
// identical arithmetic on identical input, so no pose broadcast and no second barrier. The search and the residual part of a
// keypoint run in the same wave back to back (rows_tiles' after_tile hook), so the neighbour records never change hands between
// CUs. Deterministic like the three-launch loop, with its own (fixed) summation order: per wave, per block, blocks in index order.
// A barrier that does not complete (blocks not co-resident: another process holding the CUs) times out, flags the state
// (failed = 3) and the host falls back to the three-launch loop for good.
// ================================================================================================
struct PersistShared {
    GnState state;                       // this block's copy of the solver state — identical in every block
    SolveScratch solve;
    double comb[ROW_WAVES][SYS_N];
    TieScratch tie[ROW_WAVES];
    int timeout;
};
constexpr int GN_FAILED_BARRIER = 3;     // GnState::failed: the in-kernel barrier timed out (host falls back)

template <int NB>
inline size_t persistent_kernel_smem() { return rows_kernel_smem<NB>() + sizeof(PersistShared); }


template <int NB>
__global__ __launch_bounds__(ROW_BLOCK, 2) void k_gn_persistent(MapView map, KpView kp, GnState *st_out, const double *pose_in, double tb, double te,
                                                                int init_state, GnParams prm, double *partials, DebugView dbg, int iters,
                                                                int iters_before, int kth_valid0, int rounds, int nblk, unsigned int *bar, int slot,
                                                                double *state_copy, int min_used, int final_transform, double *sys_out,
                                                                unsigned long long *times) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((blockIdx.x & 7u) != 0u) {                    // not on the working XCD: nothing to do ...
        if (blockIdx.x == 1 && tid == 0) { bar[slot ^ 1] = 0u; bar[2 + (slot ^ 1)] = 0u; }      // ... but to zero the NEXT launch's counter and XCC mask (stream order makes it safe)
        return;
    }
    const int b = (int) (blockIdx.x >> 3);
    PersistShared &P = *reinterpret_cast<PersistShared *>(smem + rows_kernel_smem<NB>());
    // ---- prologue: every block builds the same state
    if (tid == 0) {
        if (init_state) {                             // what k_state_init does (pose normalisation ct_icp.cpp:716-717 + slerp constants)
            GnState &S = P.state;
            const Quat qb = quat_normalized(Quat{pose_in[0], pose_in[1], pose_in[2], pose_in[3]});
            const Quat qe = quat_normalized(Quat{pose_in[7], pose_in[8], pose_in[9], pose_in[10]});
            S.pose[0] = qb.x; S.pose[1] = qb.y; S.pose[2] = qb.z; S.pose[3] = qb.w;
            S.pose[7] = qe.x; S.pose[8] = qe.y; S.pose[9] = qe.z; S.pose[10] = qe.w;
            for (int c = 0; c < 3; ++c) { S.pose[4 + c] = pose_in[4 + c]; S.pose[11 + c] = pose_in[11 + c]; }
            S.tbe[0] = tb; S.tbe[1] = te;
            const SlerpPair sp = slerp_prepare(qb, qe);
            S.slerp_theta = sp.theta; S.slerp_sin = sp.sin_theta; S.slerp_linear = sp.linear; S.slerp_negate = sp.negate;
            for (int i = 0; i < 12; ++i) S.x[i] = 0.0;
            S.step_norm = 0.0;
            S.iter = 0; S.done = 0; S.failed = 0; S.n_used = 0;
            for (int i = 0; i < 4; ++i) S.solve_cycles[i] = 0ull;
            S.clk_iter_start = 0ull; S.ticks_neighborhood = 0ull; S.ticks_solve = 0ull; S.ticks_iter = 0ull;
        } else {
            P.state = *st_out;                        // a running solve: written by an earlier launch on this stream
        }
        P.timeout = 0;
    }
    __syncthreads();
    const int ntiles = (kp.n + 4 * rounds - 1) / (4 * rounds);
    const unsigned int my_xcc = (unsigned int) __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;      // HW_REG_XCC_ID[3:0]
    int it = 0;
    for (; it < iters; ++it) {
        if (P.state.done) break;                      // the same decision in every block
        if (tid == 0) P.state.clk_iter_start = wall_clock64();
        KpView kv = kp;
        kv.kth_valid = it > 0 ? 1 : kth_valid0;
        d4_t accm = {0.0, 0.0, 0.0, 0.0};
        int n_used_wave = 0;
        WaveScratch<((2 * NB + 1) * (2 * NB + 1) * (2 * NB + 1) + 3) & ~3> &W =
            reinterpret_cast<WaveScratch<((2 * NB + 1) * (2 * NB + 1) * (2 * NB + 1) + 3) & ~3> *>(smem)[wave];
        // search of a tile, then — same wave, keypoints still named by W.id — its residual part: lane l takes keypoint W.id[l]
        rows_tiles<NB, true, false, false, false, false>(map, kv, &P.state, prm, dbg, (iters_before + it) == 0 ? 1 : 0, rounds, nullptr, 0, smem,
                                           b * ROW_WAVES + wave, ntiles, nblk * ROW_WAVES, [&](int) {
            const int id = W.id[lane];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the wave's own records and world points have left for L2
            residual_tile(map, kv, &P.state, prm, dbg, 0, id < 0 ? kv.n : id, lane, W.rec, accm, n_used_wave, P.tie[wave]);
        });
        unpack_wave_sums(lane, accm, n_used_wave, P.comb[wave]);
        __syncthreads();
        // ---- publish this block's packed sums write-through, drain, arrive
        for (int e = tid; e < SYS_N; e += ROW_BLOCK) {
            double sum = 0.0;
            for (int w = 0; w < ROW_WAVES; ++w) sum += P.comb[w][e];
            partials[(size_t) b * SYS_N + e] = sum;           // plain: write-through L1, kept in this XCD's L2. Block-major here (the
                                                              // three-launch loop stores entry-major, 16 KB apart: one L2 channel)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long tc0 = __builtin_readcyclecounter();
        const unsigned long long wall0 = wall_clock64();
        if (tid == 0 && times) { times[4 * b] = P.state.clk_iter_start; times[4 * b + 1] = wall0; }     // measurement hook: block timeline of the last iteration
        if (tid == 0) {
            const unsigned int target = (unsigned int) nblk * (unsigned int) (it + 1);
            if (it == 0) __hip_atomic_fetch_or(bar + 2 + slot, 1u << my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // before the arrival
            __hip_atomic_fetch_add(bar + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(bar + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 18)) { P.timeout = 1; break; }   // ~0.1 s: the other blocks are not running — give up, do not hang the GPU
            }
            // every block has arrived, so every block's XCC bit is in the mask: one bit = one L2 = the plain stores above are visible here
            const unsigned int xccs = __hip_atomic_load(bar + 2 + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (it == 0 && (xccs & (xccs - 1u)) != 0u) P.timeout = 1;
        }
        __syncthreads();
        if (tid == 0 && times) times[4 * b + 2] = wall_clock64();
        if (P.timeout) {
            if (tid == 0) { P.state.failed = GN_FAILED_BARRIER; P.state.done = 1; }
            __syncthreads();
            break;
        }
        // ---- every block: the same fixed-order sum over the blocks (two halves of the blocks in parallel, 16 loads in flight per
        // thread, then half 0 + half 1), then the same solve
        if (tid < 2 * SYS_N) {
            const int e = tid % SYS_N, half = tid / SYS_N;
            const int j0 = half == 0 ? 0 : nblk / 2, j1 = half == 0 ? nblk / 2 : nblk;
            const double *col = partials + e;                          // entry e of block j: coalesced across the lanes
            double sum = 0.0;
            int j = j0;
            for (; j + 16 <= j1; j += 16) {
                double v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = sc1_load(col + (size_t) (j + q) * SYS_N);
#pragma unroll
                for (int q = 0; q < 16; ++q) sum += v[q];
            }
            for (; j < j1; ++j) sum += sc1_load(col + (size_t) j * SYS_N);
            P.comb[half][e] = sum;
        }
        __syncthreads();
        for (int e = tid; e < SYS_N; e += ROW_BLOCK) {
            const double sum = e < SYS_USED ? P.comb[0][e] + P.comb[1][e] : 0.0;
            P.solve.sys[e] = sum;
            if (b == 0) sys_out[e] = sum;                             // the packed system of the last accumulation (ctgn_get_system)
        }
        if (tid == 0 && times) times[4 * b + 3] = wall_clock64();
        __syncthreads();
        if (wave == 0) solve_wave0(P.solve, &P.state, prm, min_used, lane, tc0, wall0);
        __syncthreads();
    }
    // ---- epilogue: the final re-transform (ct_icp.cpp:964-966 of the last executed iteration; k_transform's rule) and the state
    if (final_transform && P.state.iter > 0 && P.state.failed != GN_FAILED_BARRIER) {
        for (int i = b * ROW_BLOCK + tid; i < kp.n; i += nblk * ROW_BLOCK) {
            const Vec3 raw{kp.rx[i], kp.ry[i], kp.rz[i]};
            const double alpha = alpha_timestamp(kp.t[i], P.state.tbe[0], P.state.tbe[1]);
            const Vec3 p = ct_transform(&P.state, alpha, raw);
            kp.wx[i] = p.x; kp.wy[i] = p.y; kp.wz[i] = p.z;
        }
    }
    if (b == 0 && tid < (int) (sizeof(GnState) / 8)) {
        const double v = reinterpret_cast<const double *>(&P.state)[tid];
        reinterpret_cast<double *>(st_out)[tid] = v;
        if (state_copy) state_copy[tid] = v;
    }
}

// Re-transform every keypoint with the final pose (ct_icp.cpp:964-966 of the last executed iteration).
// state_copy (optional): the GnState is mirrored right behind the world arrays so that one device-to-host copy brings
// back both the world points and the final state.
__global__ __launch_bounds__(256) void k_transform(KpView kp, const GnState *st, double *state_copy = nullptr) {
    if (state_copy && blockIdx.x == 0 && threadIdx.x < sizeof(GnState) / 8)
        state_copy[threadIdx.x] = reinterpret_cast<const double *>(st)[threadIdx.x];
    // Nothing solved yet: the world points stay as uploaded. A soft failure at iteration i >= 1 (ct_icp.cpp:860-871) returns the world
    // points the failing iteration saw, T(pose after iteration i-1) raw — which is what this recomputes from the (unchanged) pose;
    // in ordered mode the GN kernels wrote them to the position-ordered working copy, not here.
    if (st->iter == 0) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kp.n; i += gridDim.x * blockDim.x) {
        Vec3 raw{kp.rx[i], kp.ry[i], kp.rz[i]};
        double alpha = alpha_timestamp(kp.t[i], st->tbe[0], st->tbe[1]);
        Vec3 p = ct_transform(st, alpha, raw);
        kp.wx[i] = p.x; kp.wy[i] = p.y; kp.wz[i] = p.z;
    }
}

// Full-scan undistortion (reference src/ct_icp/odometry.cpp:461-486): out = InterpolatePose(t) * raw for n points given as
// SoA arrays [x | y | z | t] with stride `cap`; pose = begin|end (14 doubles), slerp constants computed per thread block.
// sel (optional): point i of the output is point sel[i] of the input (the sampled frame of the frame pipeline); out_cap = stride
// of the output arrays.
// in_es: coordinate a of input point j is in[j * in_es + a * cap] (x y z t records: in_es 4, cap 1; then out_cap must be given).
// out_aos: the output is written as x y z records (out[3 i + a]) instead of three planes.
// out_index (with out_aos): record i of the output is written at position out_index[i] — the frame pipeline's un-shuffle (every scan point
// back in the caller's numbering on the device, so that the read-back lands in the caller's order).
__global__ __launch_bounds__(256) void k_transform_points(const double *in, double *out, int n, size_t cap, const double *pose,
                                                          double tb, double te, const uint32_t *sel = nullptr, size_t out_cap = 0,
                                                          size_t in_es = 1, int out_aos = 0, const uint32_t *out_index = nullptr) {
    if (out_cap == 0) out_cap = cap;
    __shared__ GnState s;
    if (threadIdx.x == 0) {
        const Quat qb = quat_normalized(Quat{pose[0], pose[1], pose[2], pose[3]});
        const Quat qe = quat_normalized(Quat{pose[7], pose[8], pose[9], pose[10]});
        s.pose[0] = qb.x; s.pose[1] = qb.y; s.pose[2] = qb.z; s.pose[3] = qb.w;
        s.pose[7] = qe.x; s.pose[8] = qe.y; s.pose[9] = qe.z; s.pose[10] = qe.w;
        for (int c = 0; c < 3; ++c) { s.pose[4 + c] = pose[4 + c]; s.pose[11 + c] = pose[11 + c]; }
        const SlerpPair sp = slerp_prepare(qb, qe);
        s.slerp_theta = sp.theta; s.slerp_sin = sp.sin_theta; s.slerp_linear = sp.linear; s.slerp_negate = sp.negate;
    }
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double *q = in + (sel ? (size_t) sel[i] : (size_t) i) * in_es;
        const Vec3 raw{q[0], q[cap], q[2 * cap]};
        const double alpha = alpha_timestamp(q[3 * cap], tb, te);
        const Vec3 p = ct_transform(&s, alpha, raw);
        if (out_aos) { const size_t o = out_index ? (size_t) out_index[i] : (size_t) i; out[3 * o] = p.x; out[3 * o + 1] = p.y; out[3 * o + 2] = p.z; }
        else { out[i] = p.x; out[out_cap + i] = p.y; out[2 * out_cap + i] = p.z; }
    }
}

// Frame pipeline: keypoint i = scan point sel[i] (x y z t records): raw point and timestamp into the solver's keypoint arrays
// (planes c apart).
// x y z rows + timestamps (or one timestamp for all: t == nullptr) -> the frame's x y z t records
__global__ __launch_bounds__(256) void k_frame_records(const double *xyz, const double *t, double t_all, int n, double *rec) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = xyz[3 * (size_t) i], y = xyz[3 * (size_t) i + 1], z = xyz[3 * (size_t) i + 2];
        const double ti = t ? t[i] : t_all;
        double *q = rec + 4 * (size_t) i;
        q[0] = x; q[1] = y; q[2] = z; q[3] = ti;
    }
}

// Frame pipeline, device-side shuffle (ctgn_frame_options::shuffle_seed): the reference shuffles the scan with its std::mt19937_64 before
// sub_sample_frame so that WHICH point of a voxel survives is a random choice (odometry.cpp:349); a host that does not need that very
// permutation has it made here instead — a keyed bijection of [0, n): six Feistel rounds over the next even number of bits, walked until
// the image falls inside [0, n) (cycle walking keeps it a bijection). One thread per output position: rec_out[j] = rec_in[perm(j)],
// order[j] = perm(j). No sort, no host work, one launch.
__device__ __forceinline__ uint32_t shuffle_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t shuffle_round_key(unsigned long long seed, int round) {     // splitmix64 of (seed, round)
    unsigned long long z = seed + 0x9e3779b97f4a7c15ull * (unsigned long long) (round + 1);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (uint32_t) (z ^ (z >> 31));
}
__device__ __forceinline__ uint32_t shuffle_perm(uint32_t j, uint32_t n, int half_bits, unsigned long long seed) {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t key[6];
#pragma unroll
    for (int round = 0; round < 6; ++round) key[round] = shuffle_round_key(seed, round);
    uint32_t x = j;
    do {
        uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
        for (int round = 0; round < 6; ++round) {
            const uint32_t t = l ^ (shuffle_mix(r ^ key[round]) & mask);
            l = r; r = t;
        }
        x = (l << half_bits) | r;
    } while (x >= n);
    return x;
}
// order_in == nullptr: order[j] = the keyed permutation of j (written to order_out). order_in != nullptr: the CALLER's processing order;
// an index out of range or met twice sets *bad (seen: n zeroed words).
__global__ __launch_bounds__(256) void k_frame_permute(const double *rec_in, double *rec_out, const uint32_t *order_in, uint32_t *order_out, int n,
                                                       int half_bits, unsigned long long seed, unsigned int *seen, int *bad) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        uint32_t i;
        if (order_in) {
            i = order_in[j];
            if (i >= (uint32_t) n || atomicExch(&seen[i], 1u) != 0u) { atomicOr(bad, 1); continue; }
        } else {
            i = shuffle_perm((uint32_t) j, (uint32_t) n, half_bits, seed);
            order_out[j] = i;
        }
        const double *q = rec_in + 4 * (size_t) i;
        double *o = rec_out + 4 * (size_t) j;
        o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
    }
}

// positions in processing order -> the caller's point numbers (order[j] = caller index of the point at position j)
__global__ __launch_bounds__(256) void k_frame_translate(const uint32_t *sel, const uint32_t *order, int n, uint32_t *out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = order[sel[i]];
}

__global__ __launch_bounds__(256) void k_frame_keypoints(const double *scan, const uint32_t *sel, int n, double *kp, size_t c) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double *q = scan + (size_t) sel[i] * 4;
        kp[i] = q[0]; kp[c + i] = q[1]; kp[2 * c + i] = q[2]; kp[3 * c + i] = q[3];
    }
}

// GnState initialisation on the device (pose normalisation :716-717 + slerp constants).
__global__ void k_state_init(GnState *st, const double *pose_in, double tb, double te) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    state_init_body(st, pose_in, tb, te);
}

// ================================================================================================
// strided views living in device memory (ctgn_view.base may be a device pointer): gather into / scatter from the
// handle's SoA arrays without touching the host
// ================================================================================================
__global__ __launch_bounds__(256) void k_view_gather(const char *base, size_t stride, int is_f64, int ncomp, double *dst,
                                                      size_t dst_stride, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const char *p = base + (size_t) i * stride;
        for (int c = 0; c < ncomp; ++c)
            dst[(size_t) c * dst_stride + i] = is_f64 ? reinterpret_cast<const double *>(p)[c] : (double) reinterpret_cast<const float *>(p)[c];
    }
}

__global__ __launch_bounds__(256) void k_view_scatter(const double *src, size_t src_stride, int ncomp, char *base, size_t stride,
                                                       int is_f64, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        char *p = base + (size_t) i * stride;
        for (int c = 0; c < ncomp; ++c) {
            const double v = src[(size_t) c * src_stride + i];
            if (is_f64) reinterpret_cast<double *>(p)[c] = v;
            else reinterpret_cast<float *>(p)[c] = (float) v;
        }
    }
}

// out[0] = min, out[1] = max of t[0..n); NaN anywhere makes out[1] NaN (the host check then fails, as for host views)
__global__ __launch_bounds__(1024) void k_minmax(const double *t, int n, double *out) {
    __shared__ double s_lo[16], s_hi[16];
    __shared__ int s_nan[16];
    double lo = INFINITY, hi = -INFINITY;
    int has_nan = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = t[i];
        if (v != v) has_nan = 1;
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, d)); hi = fmax(hi, __shfl_xor(hi, d)); has_nan |= __shfl_xor(has_nan, d);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_nan[wave] = has_nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int) (blockDim.x >> 6); ++w) { lo = fmin(lo, s_lo[w]); hi = fmax(hi, s_hi[w]); has_nan |= s_nan[w]; }
        out[0] = lo;
        out[1] = has_nan ? NAN : hi;
    }
}

// ================================================================================================
// map maintenance + queries
// ================================================================================================
__global__ void k_scatter_slots(Slot *slots, const SlotEdit *edits, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[edits[i].slot] = edits[i].value;     // edits are in log order; duplicates carry increasing
}                                                         // counts and are resolved on the host (last one wins)

__global__ void k_scatter_points(double *blocks, int blk, const PointEdit *edits, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        PointEdit e = edits[i];
        double *bx = blocks + (size_t) e.block * 3 * blk;
        bx[3 * e.index] = e.x; bx[3 * e.index + 1] = e.y; bx[3 * e.index + 2] = e.z;
    }
}

// RadiusSearch for a batch of queries (map.h:449-514): farthest-first output, as the reference drains its heap.
__global__ __launch_bounds__(LANE_BLOCK) void k_radius_search(MapView map, const double *queries, int n, int k,
                                                              double *out_xyz, int *out_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    double *d2s = reinterpret_cast<double *>(smem);
    uint32_t *ids = reinterpret_cast<uint32_t *>(d2s + (size_t) KMAX * LANE_BLOCK);
    const int i = blockIdx.x * LANE_BLOCK + tid;
    if (i >= n) return;
    Vec3 p{queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]};
    int cnt = lane_search(map, p, k, d2s + tid, ids + tid, LANE_BLOCK, nullptr);
    out_count[i] = cnt;
    for (int j = 0; j < cnt; ++j) {
        Vec3 c = map_point(map, ids[(cnt - 1 - j) * LANE_BLOCK + tid]);
        double *o = out_xyz + ((size_t) i * k + j) * 3;
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
    }
}

// Counting pass for the roofline: voxels probed / hit, points scanned, at the current world points.
__global__ __launch_bounds__(256) void k_count_traffic(MapView map, KpView kp, Counters *out) {
    unsigned long long c_probe = 0, c_hit = 0, c_pts = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kp.n; i += gridDim.x * blockDim.x) {
        int kx = voxel_coord(kp.wx[i], map.resolution), ky = voxel_coord(kp.wy[i], map.resolution),
            kz = voxel_coord(kp.wz[i], map.resolution);
        if (!(sweep_in_short_range(kx, map.nb) && sweep_in_short_range(ky, map.nb) && sweep_in_short_range(kz, map.nb)))
            continue;
        for (int vx = kx - map.nb; vx <= kx + map.nb; ++vx)
            for (int vy = ky - map.nb; vy <= ky + map.nb; ++vy)
                for (int vz = kz - map.nb; vz <= kz + map.nb; ++vz) {
                    uint32_t bc = map_lookup(map, vx, vy, vz);
                    c_probe++;
                    if (bc) { c_hit++; c_pts += bc & 127u; }
                }
    }
    atomicAdd(&out->probed, c_probe);
    atomicAdd(&out->hit, c_hit);
    atomicAdd(&out->points, c_pts);
}

}  // namespace ctgn
