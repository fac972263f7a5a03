// ctgn_map.hpp — host mirror of the GPU voxel map, kept in the SAME layout as the device copy so that an
// upload is a memcpy (full) or a scatter of the logged edits (delta).
//
// Mirrors the write side of ct_icp::MultipleResolutionVoxelMap (reference include/ct_icp/map.h):
//   InsertPointInVoxelMap :261-293, RemoveElementsFarFromLocation :305-322, ClearMap :296, NumPoints :341-347,
//   SearchParamsFromRadiusSearch :416-432.
//
// Layout of one resolution level
//   slots  : open-addressing table, capacity 2^p, one 16-byte slot per entry
//              { u64 key (3 x 21-bit biased voxel coords) ; u32 block ; u32 count }
//            key == EMPTY terminates a probe, key == TOMB (evicted voxel) does not.
//   blocks : fixed-capacity point blocks of 3 * BLK doubles (BLK = max_num_points of the level), points in insertion order, ARRAY OF
//            POINTS inside the block since round 3: x0 y0 z0 | x1 y1 z1 | ... A point is 24 contiguous bytes = one or two 64-byte
//            lines, where the earlier x[BLK] | y[BLK] | z[BLK] planes spread it over three lines 8 * BLK bytes apart: the residual
//            kernel's 20 gathers per keypoint touch a third of the lines, and a 16-lane row of the search kernel fetches 16 points
//            with two loads (x, y as 16 bytes + z) instead of three, over the same 384 bytes.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ctgn_math.hpp"      // sq_norm3: the reference build's rounding of a squared distance

namespace ctgn {

struct Slot {
    uint64_t key;
    uint32_t block;
    uint32_t count;
};
static_assert(sizeof(Slot) == 16, "slot must be 16 bytes");

constexpr uint64_t KEY_EMPTY = ~0ull;
constexpr uint64_t KEY_TOMB = ~0ull - 1;
constexpr int COORD_BIAS = 1 << 20;
constexpr int COORD_LIMIT = (1 << 20) - 2;

__host__ __device__ inline uint64_t pack_key(int x, int y, int z) {
    return (uint64_t) (uint32_t) (x + COORD_BIAS) | ((uint64_t) (uint32_t) (y + COORD_BIAS) << 21) |
           ((uint64_t) (uint32_t) (z + COORD_BIAS) << 42);
}

__host__ __device__ inline uint32_t hash_key(uint64_t k, uint32_t mask) {
    k ^= k >> 31;
    k *= 0x9E3779B97F4A7C15ull;
    k ^= k >> 29;
    return (uint32_t) (k >> 16) & mask;
}

// Voxel::Coordinates (reference src/SlamCore/types.cxx:13-20): int(p / voxel_size), truncation toward zero.
__host__ __device__ inline int voxel_coord(double p, double voxel_size) { return (int) (p / voxel_size); }

struct PointEdit {      // a point appended to a block since the last device sync
    uint32_t block;
    uint32_t index;
    double x, y, z;
};
struct SlotEdit {       // a slot whose value changed since the last device sync
    uint32_t slot;
    uint32_t _pad;
    Slot value;
};

struct VoxelLevel {
    double resolution = 0.5;
    double min_distance = 0.1;
    int blk = 40;                          // max_num_points

    std::vector<Slot> slots;               // capacity = mask + 1
    uint32_t mask = 0;
    std::vector<double> blocks;            // nblocks_cap * 3 * blk
    uint32_t nblocks_cap = 0;
    uint32_t nblocks_used = 0;             // high-water mark
    std::vector<uint32_t> free_blocks;
    uint64_t num_voxels = 0, num_tombs = 0, num_points = 0;

    // edit log for the delta upload
    bool log_edits = false;                // true once a device copy exists
    bool need_full_upload = true;
    std::vector<PointEdit> point_edits;
    std::vector<SlotEdit> slot_edits;

    void init(double res, double min_dist, int max_pts, uint64_t initial_voxels) {
        resolution = res;
        min_distance = min_dist;
        blk = max_pts < 1 ? 1 : max_pts;
        uint64_t cap = 1024;
        while (cap < 8 * initial_voxels) cap <<= 1;
        slots.assign(cap, Slot{KEY_EMPTY, 0, 0});
        mask = (uint32_t) (cap - 1);
        nblocks_cap = (uint32_t) (initial_voxels > 256 ? initial_voxels : 256);
        blocks.assign((size_t) nblocks_cap * 3 * blk, 0.0);
        nblocks_used = 0;
        free_blocks.clear();
        num_voxels = num_tombs = num_points = 0;
        need_full_upload = true;
        point_edits.clear();
        slot_edits.clear();
    }

    void clear() { init(resolution, min_distance, blk, 0); }

    inline double *bx(uint32_t b) { return &blocks[(size_t) b * 3 * blk]; }
    inline const double *bx(uint32_t b) const { return &blocks[(size_t) b * 3 * blk]; }

    void log_slot(uint32_t s) {
        if (log_edits && !need_full_upload) slot_edits.push_back(SlotEdit{s, 0, slots[s]});
    }

    void rehash(uint64_t new_cap) {
        std::vector<Slot> old;
        old.swap(slots);
        slots.assign(new_cap, Slot{KEY_EMPTY, 0, 0});
        mask = (uint32_t) (new_cap - 1);
        num_tombs = 0;
        for (const Slot &s : old) {
            if (s.key == KEY_EMPTY || s.key == KEY_TOMB) continue;
            uint32_t i = hash_key(s.key, mask);
            while (slots[i].key != KEY_EMPTY) i = (i + 1) & mask;
            slots[i] = s;
        }
        need_full_upload = true;
        slot_edits.clear();
        point_edits.clear();
    }

    uint32_t alloc_block() {
        if (!free_blocks.empty()) {
            uint32_t b = free_blocks.back();
            free_blocks.pop_back();
            return b;
        }
        if (nblocks_used == nblocks_cap) {
            nblocks_cap = nblocks_cap * 2;
            blocks.resize((size_t) nblocks_cap * 3 * blk, 0.0);
            need_full_upload = true;
            slot_edits.clear();
            point_edits.clear();
        }
        return nblocks_used++;
    }

    // returns the slot index of `key`, or -1
    int64_t find(uint64_t key) const {
        uint32_t i = hash_key(key, mask);
        for (;;) {
            const Slot &s = slots[i];
            if (s.key == key) return i;
            if (s.key == KEY_EMPTY) return -1;
            i = (i + 1) & mask;
        }
    }

    // InsertPointInVoxelMap (map.h:261-293). Returns 1 if inserted, 0 if dropped, -1 if out of key range.
    int insert_point(double px, double py, double pz) {
        int vx = voxel_coord(px, resolution), vy = voxel_coord(py, resolution), vz = voxel_coord(pz, resolution);
        if (vx < -COORD_LIMIT || vx > COORD_LIMIT || vy < -COORD_LIMIT || vy > COORD_LIMIT || vz < -COORD_LIMIT ||
            vz > COORD_LIMIT || !std::isfinite(px) || !std::isfinite(py) || !std::isfinite(pz))
            return -1;
        uint64_t key = pack_key(vx, vy, vz);
        uint32_t i = hash_key(key, mask);
        int64_t first_tomb = -1;
        for (;;) {
            Slot &s = slots[i];
            if (s.key == key) break;
            if (s.key == KEY_EMPTY) { i = (first_tomb >= 0) ? (uint32_t) first_tomb : i; goto new_voxel; }
            if (s.key == KEY_TOMB && first_tomb < 0) first_tomb = i;
            i = (i + 1) & mask;
        }
        {   // existing voxel: map.h:275-291
            Slot &s = slots[i];
            if ((int) s.count < blk) {
                const double *x = bx(s.block);
                double sq_min = 1.7976931348623157e308;
                for (uint32_t k = 0; k < s.count; ++k) {
                    double dx = x[3 * k] - px, dy = x[3 * k + 1] - py, dz = x[3 * k + 2] - pz;
                    double sq = sq_norm3(dx, dy, dz);
                    if (sq < sq_min) sq_min = sq;
                }
                if (sq_min > min_distance * min_distance) {
                    double *wx = bx(s.block);
                    wx[3 * s.count] = px; wx[3 * s.count + 1] = py; wx[3 * s.count + 2] = pz;
                    if (log_edits && !need_full_upload) point_edits.push_back(PointEdit{s.block, s.count, px, py, pz});
                    s.count++;
                    num_points++;
                    log_slot(i);
                    return 1;
                }
            }
            return 0;
        }
    new_voxel:
        {   // map.h:267-273
            if (slots[i].key == KEY_TOMB) num_tombs--;
            uint32_t b = alloc_block();
            double *wx = bx(b);
            wx[0] = px; wx[1] = py; wx[2] = pz;
            slots[i] = Slot{key, b, 1};
            if (log_edits && !need_full_upload) point_edits.push_back(PointEdit{b, 0, px, py, pz});
            log_slot(i);
            num_voxels++;
            num_points++;
            // load factor (live + tombstones) kept in [1/8, 1/4]: a miss then ends after ~1.2 probes on average,
            // which matters because a wave waits for the slowest of its 64 concurrent probes
            if ((num_voxels + num_tombs) * 4 > (uint64_t) mask + 1) {
                uint64_t cap = 1024;
                while (cap < 8 * num_voxels) cap <<= 1;
                rehash(cap);
            }
            return 1;
        }
    }

    // RemoveElementsFarFromLocation (map.h:305-322): voxel removed iff ||first point - location|| > distance.
    void remove_far(const double loc[3], double distance) {
        for (uint32_t i = 0; i <= mask; ++i) {
            Slot &s = slots[i];
            if (s.key == KEY_EMPTY || s.key == KEY_TOMB) continue;
            const double *x = bx(s.block);
            double dx = x[0] - loc[0], dy = x[1] - loc[1], dz = x[2] - loc[2];
            if (std::sqrt(sq_norm3(dx, dy, dz)) > distance) {
                num_points -= s.count;
                num_voxels--;
                num_tombs++;
                free_blocks.push_back(s.block);
                s = Slot{KEY_TOMB, 0, 0};
                log_slot(i);
            }
        }
    }

    uint64_t export_points(double *out, uint64_t cap) const {
        uint64_t k = 0;
        for (uint32_t i = 0; i <= mask; ++i) {
            const Slot &s = slots[i];
            if (s.key == KEY_EMPTY || s.key == KEY_TOMB) continue;
            const double *x = bx(s.block);
            for (uint32_t j = 0; j < s.count; ++j, ++k)
                if (out && k < cap) { out[3 * k] = x[3 * j]; out[3 * k + 1] = x[3 * j + 1]; out[3 * k + 2] = x[3 * j + 2]; }
        }
        return k;
    }
};

// SearchParamsFromRadiusSearch (map.h:416-432)
inline void search_params(const std::vector<VoxelLevel> &levels, double radius, int *map_id, double *resolution,
                          int *nb) {
    int it = 0;
    while (it < (int) levels.size() && levels[it].resolution <= radius) ++it;
    int idx = it - 1 < 0 ? 0 : it - 1;
    *map_id = idx;
    *resolution = levels[idx].resolution;
    *nb = (int) std::ceil(radius / levels[idx].resolution);
}

// Largest double t with sqrt(t) <= radius under correctly rounded sqrt: `sqrt(d2) > radius` (the reference's
// test on the norm, map.h:491-492) is then exactly `d2 > t`, and the kernels never need a per-candidate sqrt.
inline double radius_sq_threshold(double radius) {
    if (!(radius > 0)) return 0.0;
    double t = radius * radius;
    while (std::sqrt(t) > radius) t = std::nextafter(t, 0.0);
    while (std::sqrt(std::nextafter(t, INFINITY)) <= radius) t = std::nextafter(t, INFINITY);
    return t;
}

}  // namespace ctgn
