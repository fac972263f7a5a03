// ctgn_devmap.hip — device-resident voxel-map maintenance, the samplers and the keypoint ordering (see ctgn_devmap.hpp). Sorting
// and compaction are the kernels of ctgn_sort.hpp (stable radix sort in one launch for a frame's batch, three kernels per executed
// pass beyond 16 k keys; ordered compaction without a scan kernel) — no library underneath.
#include "ctgn_devmap.hpp"
#include "ctgn_sort.hpp"

#include <algorithm>
#include <vector>

namespace ctgn {

#define DM_CHK(call)                          \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) return e_;      \
    } while (0)

// ------------------------------------------------------------------------------------------------ kernels
__global__ void k_dm_fill_slots(Slot *slots, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        slots[i] = Slot{KEY_EMPTY, 0u, 0u};
}

// voxel key of every staged point (Voxel::Coordinates, src/SlamCore/types.cxx:13-20); out-of-range / non-finite points
// get KEY_EMPTY (they sort to the end and are skipped) and raise the range flag.
__global__ void k_dm_keys(const double *pts, size_t cap, size_t n, double resolution, uint64_t *keys, uint32_t *idx,
                          DevCounters *cnt) {
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = pts[i], y = pts[cap + i], z = pts[2 * cap + i];
    uint64_t key = KEY_EMPTY;
    const bool fin = isfinite(x) && isfinite(y) && isfinite(z);
    if (fin) {
        const int vx = voxel_coord(x, resolution), vy = voxel_coord(y, resolution), vz = voxel_coord(z, resolution);
        if (vx >= -COORD_LIMIT && vx <= COORD_LIMIT && vy >= -COORD_LIMIT && vy <= COORD_LIMIT && vz >= -COORD_LIMIT && vz <= COORD_LIMIT)
            key = pack_key(vx, vy, vz);
    }
    if (key == KEY_EMPTY) atomicOr(&cnt->range_error, 1u);
    keys[i] = key;
    idx[i] = (uint32_t) i;
}

// One thread per sorted position; the thread at the head of a run of equal keys inserts the whole run, in original
// order (the sort is stable), with the reference's rule (map.h:261-293).
__global__ void k_dm_insert(Slot *slots, uint32_t mask, double *blocks, int blk, uint32_t nblocks_cap, uint32_t *free_list,
                            DevCounters *cnt, const uint64_t *keys, const uint32_t *idx, size_t n, const double *pts, size_t cap,
                            double min_dist_sq, uint8_t *inserted, const int *skip) {
    if (skip && *skip) return;                            // the frame pipeline's gate: the registration that produced the batch failed
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    if (key == KEY_EMPTY) return;
    if (i > 0 && keys[i - 1] == key) return;              // not the head of its run
    // find the voxel, or claim an EMPTY slot for it (tombstones are not reused here; the host rehashes when they pile up)
    uint32_t s = hash_key(key, mask);
    bool fresh = false;
    for (uint32_t probes = 0;; ++probes) {
        const unsigned long long cur = *reinterpret_cast<volatile unsigned long long *>(&slots[s].key);
        if (cur == key) break;
        if (cur == KEY_EMPTY) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&slots[s].key), KEY_EMPTY, key);
            if (old == KEY_EMPTY) { fresh = true; break; }
            if (old == key) break;
        }
        s = (s + 1) & mask;
        if (probes > mask) { atomicOr(&cnt->overflow, 1u); return; }
    }
    uint32_t block, count;
    if (fresh) {
        const int t = atomicSub(&cnt->free_top, 1) - 1;
        if (t >= 0) {
            block = free_list[t];
        } else {
            atomicAdd(&cnt->free_top, 1);
            block = atomicAdd(&cnt->next_block, 1u);
            if (block >= nblocks_cap) { atomicOr(&cnt->overflow, 2u); slots[s].block = 0; slots[s].count = 0; return; }
        }
        count = 0;
        atomicAdd(&cnt->num_voxels, 1ull);
    } else {
        block = slots[s].block;
        count = slots[s].count;
    }
    double *bx = blocks + (size_t) block * 3 * blk;              // points of the voxel: x y z | x y z | ... (ctgn_map.hpp)
    unsigned long long added = 0;
    for (size_t j = i; j < n && keys[j] == key; ++j) {
        const uint32_t pi = idx[j];
        const double px = pts[pi], py = pts[cap + pi], pz = pts[2 * cap + pi];
        bool take = false;
        if (count == 0) {
            take = true;                                                     // map.h:267-273 (new voxel: always)
        } else if ((int) count < blk) {                                      // map.h:275-291
            double sq_min = 1.7976931348623157e308;
            for (uint32_t q = 0; q < count; ++q) {
                const double dx = bx[3 * q] - px, dy = bx[3 * q + 1] - py, dz = bx[3 * q + 2] - pz;
                const double sq = sq_norm3(dx, dy, dz);
                if (sq < sq_min) sq_min = sq;
            }
            take = sq_min > min_dist_sq;
        }
        if (take) {
            bx[3 * count] = px; bx[3 * count + 1] = py; bx[3 * count + 2] = pz;
            ++count;
            ++added;
            inserted[pi] = 1;
        }
    }
    slots[s].block = block;
    slots[s].count = count;
    if (added) atomicAdd(&cnt->num_points, added);
}

// RemoveElementsFarFromLocation (map.h:305-322): voxel removed iff ||first point - location|| > distance.
__global__ void k_dm_remove_far(Slot *slots, uint64_t nslots, const double *blocks, int blk, uint32_t *free_list, DevCounters *cnt,
                                double lx, double ly, double lz, double distance, double resolution, const double *loc_dev,
                                const int *failed_dev = nullptr) {
    // the fused frame call enqueues this behind a registration whose outcome the host has not seen: a HARD error of that solve (GnState::failed
    // >= 3: in-kernel barrier timed out, a peer rank failed — the call returns an error) must leave the map untouched, as include/ctgn.h promises;
    // a soft failure (1: too few keypoints) evicts round the unchanged pose, as the reference's loop does
    if (failed_dev && *failed_dev >= 3) return;
    if (loc_dev) { lx = loc_dev[0]; ly = loc_dev[1]; lz = loc_dev[2]; }      // the location lives on the device (a pose the host has not seen yet)
    for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < nslots; i += (uint64_t) gridDim.x * blockDim.x) {
        const Slot s = slots[i];
        if (s.key == KEY_EMPTY || s.key == KEY_TOMB) continue;
        // The voxel's own coordinates bound its first point: int(p / res) == v puts p inside [(v - 1) res, (v + 1) res] on every axis
        // (truncation toward zero makes voxel 0 two cells wide; the one-cell margin also swallows the rounding of the division). A
        // voxel whose whole box is nearer than `distance` stays without its block being read — all but a thin shell of a map that is
        // trimmed every frame; the decision for the shell (and for far voxels) is the reference's, on the point itself.
        {
            double far2 = 0.0;
            const double l[3] = {lx, ly, lz};
            for (int a = 0; a < 3; ++a) {
                const int v = (int) ((s.key >> (21 * a)) & 0x1FFFFFu) - COORD_BIAS;
                const double lo = (double) (v - 1) * resolution - l[a], hi = (double) (v + 1) * resolution - l[a];
                const double m = fmax(fabs(lo), fabs(hi));
                far2 += m * m;
            }
            if (sqrt(far2) * (1.0 + 1e-9) < distance) continue;
        }
        const double *bx = blocks + (size_t) s.block * 3 * blk;
        const double dx = bx[0] - lx, dy = bx[1] - ly, dz = bx[2] - lz;
        if (sqrt(sq_norm3(dx, dy, dz)) > distance) {
            slots[i] = Slot{KEY_TOMB, 0u, 0u};
            const int t = atomicAdd(&cnt->free_top, 1);
            free_list[t] = s.block;
            atomicAdd(&cnt->num_tombs, 1ull);
            atomicAdd(&cnt->num_voxels, ~0ull);                   // -1
            atomicAdd(&cnt->num_points, ~((unsigned long long) s.count) + 1ull);     // -count
        }
    }
}

// rebuild into a larger (tombstone-free) table
__global__ void k_dm_rehash(const Slot *old_slots, uint64_t old_n, Slot *slots, uint32_t mask) {
    for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < old_n; i += (uint64_t) gridDim.x * blockDim.x) {
        const Slot s = old_slots[i];
        if (s.key == KEY_EMPTY || s.key == KEY_TOMB) continue;
        uint32_t d = hash_key(s.key, mask);
        for (;;) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&slots[d].key), KEY_EMPTY, s.key);
            if (old == KEY_EMPTY) break;
            d = (d + 1) & mask;
        }
        slots[d].block = s.block;
        slots[d].count = s.count;
    }
}

// grid sampling keys: static_cast<short>(p / voxel_size) per axis (ct_icp.cpp:70-72), three int16 packed in 48 bits.
// Grid sampling without a sort: every point claims its voxel's slot in a scratch hash table and lowers the slot's point index
// with atomicMin; a point survives iff it is the smallest index of its voxel, i.e. the first one the reference's loop inserts
// (ct_icp.cpp:73-75). Integer atomics only: the result does not depend on the execution order.
constexpr unsigned long long GS_EMPTY = ~0ull;
// Point i's coordinate a is pts[i * es + a * cap]: planes `cap` apart (es = 1) or x y z t records (es = 4, cap = 1).
__global__ void k_gs_hash(const double *pts, size_t cap, size_t n, double voxel_size, unsigned long long *tkeys, uint32_t *tfirst,
                          uint32_t mask, uint32_t *slot_of, const uint8_t *active, size_t es = 1) {
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (active && !active[i]) return;                 // second-level sampling: only the points the first level kept take part
    const double *p = pts + i * es;
    const uint16_t vx = (uint16_t) (short) (int) (p[0] / voxel_size), vy = (uint16_t) (short) (int) (p[cap] / voxel_size),
                   vz = (uint16_t) (short) (int) (p[2 * cap] / voxel_size);
    const unsigned long long key = (unsigned long long) vx | ((unsigned long long) vy << 16) | ((unsigned long long) vz << 32);
    unsigned long long hsh = key * 0x9E3779B97F4A7C15ull;
    uint32_t s = (uint32_t) (hsh >> 32) & mask;
    for (;;) {
        unsigned long long cur = *reinterpret_cast<volatile unsigned long long *>(&tkeys[s]);
        if (cur == GS_EMPTY) cur = atomicCAS(&tkeys[s], GS_EMPTY, key);
        if (cur == GS_EMPTY || cur == key) break;
        s = (s + 1) & mask;
    }
    // a scan puts ~17 points into a frame voxel: once a smaller index sits in the slot the atomic is skipped (the value only falls)
    if (*reinterpret_cast<volatile uint32_t *>(&tfirst[s]) > (uint32_t) i) atomicMin(&tfirst[s], (uint32_t) i);
    slot_of[i] = s;
}

// both levels' tables in one launch
__global__ void k_gs_clear(unsigned long long *tkeys, uint32_t *tfirst, size_t nslots) {
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < nslots; i += (size_t) gridDim.x * blockDim.x) {
        tkeys[i] = GS_EMPTY;
        tfirst[i] = 0xFFFFFFFFu;
    }
}

// flags of one level + how many of them each 256-point block holds (the compaction's input). tfirst == nullptr: every (active) point.
__global__ __launch_bounds__(256) void k_gs_flags_count(const uint32_t *tfirst, const uint32_t *slot_of, size_t n, uint8_t *flags,
                                                        const uint8_t *active, uint32_t *block_counts) {
    __shared__ uint32_t s_cnt[4];
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    bool f = false;
    if (i < n) {
        f = (!active || active[i]) && (!tfirst || tfirst[slot_of[i]] == (uint32_t) i);
        flags[i] = f ? 1 : 0;
    }
    const uint32_t c = (uint32_t) __popcll(ballot64(f));
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// Ordered compaction of two flag arrays at once: sel_a / sel_b receive the ascending indices whose flag is set, totals[0 / 1] the
// counts. Every block adds up the counts of the blocks before it (a few hundred values out of L2) instead of waiting on a scan.
__global__ __launch_bounds__(256) void k_gs_compact2(const uint8_t *flag_a, const uint8_t *flag_b, const uint32_t *counts_a,
                                                     const uint32_t *counts_b, size_t n, uint32_t *sel_a, uint32_t *sel_b, int *totals) {
    __shared__ uint32_t s_red[2][4], s_wave[2][4];
    uint32_t pa = 0, pb = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256) { pa += counts_a[b]; pb += counts_b[b]; }
    for (int d = 32; d >= 1; d >>= 1) { pa += __shfl_xor(pa, d); pb += __shfl_xor(pb, d); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_red[0][wave] = pa; s_red[1][wave] = pb; }
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    const bool fa = i < n && flag_a[i], fb = i < n && flag_b[i];
    const unsigned long long ma = ballot64(fa), mb = ballot64(fb);
    if (lane == 0) { s_wave[0][wave] = (uint32_t) __popcll(ma); s_wave[1][wave] = (uint32_t) __popcll(mb); }
    __syncthreads();
    uint32_t base_a = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3], base_b = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    for (int w = 0; w < wave; ++w) { base_a += s_wave[0][w]; base_b += s_wave[1][w]; }
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (fa) sel_a[base_a + (uint32_t) __popcll(ma & below)] = (uint32_t) i;
    if (fb) sel_b[base_b + (uint32_t) __popcll(mb & below)] = (uint32_t) i;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {     // the last thread of the last block sees both totals
        totals[0] = (int) (base_a + (uint32_t) __popcll(ma & below) + (fa ? 1u : 0u));
        totals[1] = (int) (base_b + (uint32_t) __popcll(mb & below) + (fb ? 1u : 0u));
    }
}
__global__ void k_gs_first_flags(const uint32_t *tfirst, const uint32_t *slot_of, size_t n, uint8_t *flags, const uint8_t *active) {
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i < n) flags[i] = ((!active || active[i]) && tfirst[slot_of[i]] == (uint32_t) i) ? 1 : 0;
}

// Adaptive sampling keys (sampling.h:65-79): band from the range |p| (std::lower_bound on the distance list, minus one),
// voxel = int(p / size[band]) per axis; band in bits 60.., z, y, x biased by 2^19 in 20 bits each. Points outside
// [distance[0], distance[last]) — and |p| == distance[0], where the reference indexes entry -1 — get the invalid key and
// sort to the end. The explicit _rn intrinsics keep the compiler from contracting x*x + y*y + z*z into FMAs, so the range
// rounds like the host's.
constexpr uint64_t AS_INVALID = ~0ull;
__global__ void k_as_keys(const double *pts, size_t cap, size_t n, AdaptiveBands bands, uint64_t *keys, uint32_t *idx) {
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = pts[i], y = pts[cap + i], z = pts[2 * cap + i];
    const double d = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)), __dmul_rn(z, z)));
    int lw = 0;
    while (lw < bands.num_bands && bands.distance[lw] < d) ++lw;
    uint64_t key = AS_INVALID;
    if (d >= bands.distance[0] && d < bands.distance[bands.num_bands - 1] && lw >= 1) {
        const int band = lw - 1;
        const double sz = bands.voxel_size[band];
        const int vx = (int) (x / sz), vy = (int) (y / sz), vz = (int) (z / sz);
        key = ((uint64_t) band << 60) | ((uint64_t) (uint32_t) (vz + (1 << 19)) << 40) | ((uint64_t) (uint32_t) (vy + (1 << 19)) << 20) |
              (uint64_t) (uint32_t) (vx + (1 << 19));
    }
    keys[i] = key;
    idx[i] = (uint32_t) i;
}
// sorted keys (stable: indices ascend inside a run): keep the first k positions of every run
__global__ void k_as_flags(const uint64_t *keys, size_t n, uint32_t k, uint8_t *flags) {
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    flags[i] = (key != AS_INVALID && (i < k || keys[i - k] != key)) ? 1 : 0;
}

// home-voxel sort key of a keypoint's world point (see OrderScratch)
__global__ void k_order_keys(const double *wx, const double *wy, const double *wz, size_t n, double resolution, uint32_t *keys,
                             uint32_t *idx) {
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t vx = (uint32_t) voxel_coord(wx[i], resolution) & 4095u, vy = (uint32_t) voxel_coord(wy[i], resolution) & 4095u,
                   vz = (uint32_t) voxel_coord(wz[i], resolution) & 255u;
    keys[i] = (vx << 20) | (vy << 8) | vz;
    idx[i] = (uint32_t) i;
}

// ------------------------------------------------------------------------------------------------ host side
static hipError_t read_counters(DevLevel &L, hipStream_t stream) {
    DM_CHK(hipMemcpyAsync(&L.host, L.counters, sizeof(DevCounters), hipMemcpyDeviceToHost, stream));
    return hipStreamSynchronize(stream);
}

static hipError_t alloc_table(Slot **slots, uint64_t cap, hipStream_t stream) {
    DM_CHK(hipMalloc(reinterpret_cast<void **>(slots), cap * sizeof(Slot)));
    hipLaunchKernelGGL(k_dm_fill_slots, dim3((unsigned) std::min<uint64_t>((cap + 255) / 256, 4096)), dim3(256), 0, stream, *slots, cap);
    return hipGetLastError();
}

hipError_t devmap_level_init(DevLevel &L, double resolution, double min_distance, int blk, hipStream_t stream) {
    L.resolution = resolution;
    L.min_distance = min_distance;
    L.blk = blk < 1 ? 1 : blk;
    L.slots_cap = 1 << 14;
    L.nblocks_cap = 2048;
    DM_CHK(alloc_table(&L.slots, L.slots_cap, stream));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&L.blocks), (size_t) L.nblocks_cap * 3 * L.blk * sizeof(double)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&L.free_list), (size_t) L.nblocks_cap * sizeof(uint32_t)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&L.counters), sizeof(DevCounters)));
    DM_CHK(hipMemsetAsync(L.counters, 0, sizeof(DevCounters), stream));
    L.host = DevCounters{};
    return hipStreamSynchronize(stream);
}

void devmap_level_free(DevLevel &L) {
    if (L.slots) (void) hipFree(L.slots);
    if (L.blocks) (void) hipFree(L.blocks);
    if (L.free_list) (void) hipFree(L.free_list);
    if (L.counters) (void) hipFree(L.counters);
    L = DevLevel{};
}

hipError_t devmap_level_clear(DevLevel &L, hipStream_t stream) {
    hipLaunchKernelGGL(k_dm_fill_slots, dim3((unsigned) std::min<uint64_t>((L.slots_cap + 255) / 256, 4096)), dim3(256), 0, stream,
                       L.slots, L.slots_cap);
    DM_CHK(hipGetLastError());
    DM_CHK(hipMemsetAsync(L.counters, 0, sizeof(DevCounters), stream));
    L.host = DevCounters{};
    return hipStreamSynchronize(stream);
}

hipError_t devmap_scratch_reserve(DevMapScratch &S, size_t n) {
    if (n <= S.cap) return hipSuccess;
    devmap_scratch_free(S);
    const size_t cap = n + n / 4 + 1024;
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.pts), cap * 3 * sizeof(double)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.keys), cap * sizeof(uint64_t)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.keys_alt), cap * sizeof(uint64_t)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.idx), cap * sizeof(uint32_t)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.idx_alt), cap * sizeof(uint32_t)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.inserted), cap));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.sel_out), cap * sizeof(uint32_t)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.sel_count), sizeof(int)));
    DM_CHK(hipHostMalloc(reinterpret_cast<void **>(&S.h_pts), cap * 3 * sizeof(double), hipHostMallocDefault));
    DM_CHK(hipHostMalloc(reinterpret_cast<void **>(&S.h_inserted), cap, hipHostMallocDefault));
    DM_CHK(hipHostMalloc(reinterpret_cast<void **>(&S.h_count), sizeof(int), hipHostMallocDefault));
    size_t gs_cap = 1024;
    while (gs_cap < 2 * cap) gs_cap <<= 1;
    gs_cap *= 2;                                     // two tables: both sampling levels of a frame (devmap_frame_sampling)
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.gs_keys), gs_cap * sizeof(unsigned long long)));
    DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.gs_first), gs_cap * sizeof(uint32_t)));
    S.gs_cap = gs_cap;
    DM_CHK(sort_scratch_reserve(S.sort, cap));
    S.cap = cap;
    return hipSuccess;
}

void devmap_scratch_free(DevMapScratch &S) {
    if (S.pts) (void) hipFree(S.pts);
    if (S.keys) (void) hipFree(S.keys);
    if (S.keys_alt) (void) hipFree(S.keys_alt);
    if (S.idx) (void) hipFree(S.idx);
    if (S.idx_alt) (void) hipFree(S.idx_alt);
    if (S.inserted) (void) hipFree(S.inserted);
    if (S.sel_out) (void) hipFree(S.sel_out);
    if (S.sel_count) (void) hipFree(S.sel_count);
    sort_scratch_free(S.sort);
    if (S.h_pts) (void) hipHostFree(S.h_pts);
    if (S.h_inserted) (void) hipHostFree(S.h_inserted);
    if (S.h_count) (void) hipHostFree(S.h_count);
    if (S.gs_keys) (void) hipFree(S.gs_keys);
    if (S.gs_first) (void) hipFree(S.gs_first);
    S = DevMapScratch{};
}

// make room for up to n new voxels: load factor (live + tombstones) <= 1/4 after the batch, one block per new voxel
static hipError_t ensure_capacity(DevLevel &L, size_t n, hipStream_t stream) {
    const uint64_t worst = L.host.num_voxels + L.host.num_tombs + n;
    if (worst * 4 > L.slots_cap) {
        uint64_t cap = L.slots_cap;
        while (cap < 8 * (L.host.num_voxels + n)) cap <<= 1;
        Slot *fresh = nullptr;
        DM_CHK(alloc_table(&fresh, cap, stream));
        hipLaunchKernelGGL(k_dm_rehash, dim3((unsigned) std::min<uint64_t>((L.slots_cap + 255) / 256, 4096)), dim3(256), 0, stream,
                           L.slots, L.slots_cap, fresh, (uint32_t) (cap - 1));
        DM_CHK(hipGetLastError());
        DM_CHK(hipMemsetAsync(&L.counters->num_tombs, 0, sizeof(unsigned long long), stream));
        DM_CHK(hipStreamSynchronize(stream));
        DM_CHK(hipFree(L.slots));
        L.slots = fresh;
        L.slots_cap = cap;
        L.host.num_tombs = 0;
    }
    const uint64_t free_blocks = (uint64_t) std::max(L.host.free_top, 0);
    if ((uint64_t) L.host.next_block + n > (uint64_t) L.nblocks_cap + free_blocks) {
        uint64_t want = std::max<uint64_t>((uint64_t) L.nblocks_cap * 2, (uint64_t) L.host.next_block + n);
        if (want * 3 * L.blk * sizeof(double) >= ((uint64_t) 1 << 32)) want = (((uint64_t) 1 << 32) - 1) / (3ull * L.blk * sizeof(double));
        if (want < (uint64_t) L.host.next_block + n - free_blocks) return hipErrorOutOfMemory;     // 32-bit block offsets
        double *nb = nullptr;
        uint32_t *nf = nullptr;
        DM_CHK(hipMalloc(reinterpret_cast<void **>(&nb), want * 3 * L.blk * sizeof(double)));
        DM_CHK(hipMalloc(reinterpret_cast<void **>(&nf), want * sizeof(uint32_t)));
        // the WHOLE old pool and free list, not just what the last counter read-back knew of: the frame pipeline plans an insertion
        // while the eviction in front of it (which pushes onto the free list) is still in flight, its counters unread
        DM_CHK(hipMemcpyAsync(nb, L.blocks, (size_t) L.nblocks_cap * 3 * L.blk * sizeof(double), hipMemcpyDeviceToDevice, stream));
        DM_CHK(hipMemcpyAsync(nf, L.free_list, (size_t) L.nblocks_cap * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        DM_CHK(hipStreamSynchronize(stream));
        DM_CHK(hipFree(L.blocks));
        DM_CHK(hipFree(L.free_list));
        L.blocks = nb;
        L.free_list = nf;
        L.nblocks_cap = (uint32_t) want;
    }
    return hipSuccess;
}

hipError_t devmap_level_insert(DevLevel &L, DevMapScratch &S, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    DM_CHK(devmap_level_insert_enqueue(L, S, n, nullptr, stream));
    return read_counters(L, stream);
}

hipError_t devmap_level_read_counters(DevLevel &L, hipStream_t stream) { return read_counters(L, stream); }

// The kernels of an insertion without the counter read-back (the caller reads them later: devmap_level_read_counters). Capacity is
// planned from the counters as last read — every removal since then only freed room. `skip` (device, may be null): non-zero = leave the map alone.
hipError_t devmap_level_insert_enqueue(DevLevel &L, DevMapScratch &S, size_t n, const int *skip, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    DM_CHK(ensure_capacity(L, n, stream));
    const unsigned grid = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(k_dm_keys, dim3(grid), dim3(256), 0, stream, S.pts, S.stride, n, L.resolution, S.keys, S.idx, L.counters);
    DM_CHK(hipGetLastError());
    // stable: the points of a voxel stay in batch order, which is what the insert rule walks (map.h:261-293)
    DM_CHK(sort_pairs<uint64_t>(S.sort, S.keys, S.keys_alt, S.idx, S.idx_alt, n, 64, true, stream));
    hipLaunchKernelGGL(k_dm_insert, dim3(grid), dim3(256), 0, stream, L.slots, (uint32_t) (L.slots_cap - 1), L.blocks, L.blk, L.nblocks_cap,
                       L.free_list, L.counters, S.keys_alt, S.idx_alt, n, S.pts, S.stride, L.min_distance * L.min_distance, S.inserted, skip);
    return hipGetLastError();
}

hipError_t devmap_grid_sampling(DevMapScratch &S, size_t n, double voxel_size, uint32_t *out_idx_host, size_t *out_count,
                                hipStream_t stream) {
    *out_count = 0;
    if (n == 0) return hipSuccess;
    const unsigned grid = (unsigned) ((n + 255) / 256);
    size_t tcap = 1024;                                  // >= 2 slots per point: short probe sequences
    while (tcap < 2 * n) tcap <<= 1;
    if (tcap > S.gs_cap) return hipErrorInvalidValue;
    DM_CHK(hipMemsetAsync(S.gs_keys, 0xFF, tcap * sizeof(unsigned long long), stream));
    DM_CHK(hipMemsetAsync(S.gs_first, 0xFF, tcap * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_gs_hash, dim3(grid), dim3(256), 0, stream, S.pts, S.stride, n, voxel_size, S.gs_keys, S.gs_first, (uint32_t) (tcap - 1),
                       S.idx, (const uint8_t *) nullptr);
    hipLaunchKernelGGL(k_gs_first_flags, dim3(grid), dim3(256), 0, stream, S.gs_first, S.idx, n, S.inserted, (const uint8_t *) nullptr);
    DM_CHK(hipGetLastError());
    DM_CHK(compact_flags(S.inserted, nullptr, n, S.idx_alt, S.sel_out, S.sel_count, stream));      // ascending indices of the kept points
    DM_CHK(hipMemcpyAsync(S.h_count, S.sel_count, sizeof(int), hipMemcpyDeviceToHost, stream));
    DM_CHK(hipStreamSynchronize(stream));
    const int count = *S.h_count;
    DM_CHK(hipMemcpyAsync(out_idx_host, S.sel_out, (size_t) count * sizeof(uint32_t), hipMemcpyDefault, stream));   // host or device
    DM_CHK(hipStreamSynchronize(stream));
    *out_count = (size_t) count;
    return hipSuccess;
}

hipError_t devmap_frame_sampling(DevMapScratch &S, const double *pts, size_t stride, size_t es, size_t n, double frame_voxel,
                                 double keypoint_voxel, uint8_t *flag1, uint8_t *flag2, uint32_t *sel1, uint32_t *sel2, int *counts,
                                 hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(counts, 0, 2 * sizeof(int), stream);
    const unsigned grid = (unsigned) ((n + 255) / 256);
    size_t tcap = 1024;
    while (tcap < 2 * n) tcap <<= 1;
    if (2 * tcap > S.gs_cap || (size_t) grid * 2 > S.cap) return hipErrorInvalidValue;
    const uint32_t mask = (uint32_t) (tcap - 1);
    uint32_t *counts1 = S.sel_out, *counts2 = S.sel_out + grid;                    // per-block counts of the two levels
    hipLaunchKernelGGL(k_gs_clear, dim3((unsigned) std::min<size_t>((2 * tcap + 255) / 256, 2048)), dim3(256), 0, stream, S.gs_keys, S.gs_first,
                       2 * tcap);
    if (frame_voxel > 0) {
        hipLaunchKernelGGL(k_gs_hash, dim3(grid), dim3(256), 0, stream, pts, stride, n, frame_voxel, S.gs_keys, S.gs_first, mask, S.idx,
                           (const uint8_t *) nullptr, es);
        hipLaunchKernelGGL(k_gs_flags_count, dim3(grid), dim3(256), 0, stream, (const uint32_t *) S.gs_first, (const uint32_t *) S.idx, n, flag1,
                           (const uint8_t *) nullptr, counts1);
    } else {
        hipLaunchKernelGGL(k_gs_flags_count, dim3(grid), dim3(256), 0, stream, (const uint32_t *) nullptr, (const uint32_t *) nullptr, n, flag1,
                           (const uint8_t *) nullptr, counts1);
    }
    const uint8_t *kp_flags = flag1;
    const uint32_t *kp_counts = counts1;
    if (keypoint_voxel > 0) {
        // "first point of every voxel, in the order of the sampled frame" = smallest index among the kept points: the sampled frame is
        // the kept points in ascending index order
        hipLaunchKernelGGL(k_gs_hash, dim3(grid), dim3(256), 0, stream, pts, stride, n, keypoint_voxel, S.gs_keys + tcap, S.gs_first + tcap, mask,
                           S.idx, (const uint8_t *) flag1, es);
        hipLaunchKernelGGL(k_gs_flags_count, dim3(grid), dim3(256), 0, stream, (const uint32_t *) (S.gs_first + tcap), (const uint32_t *) S.idx, n,
                           flag2, (const uint8_t *) flag1, counts2);
        kp_flags = flag2;
        kp_counts = counts2;
    }
    hipLaunchKernelGGL(k_gs_compact2, dim3(grid), dim3(256), 0, stream, (const uint8_t *) flag1, kp_flags, (const uint32_t *) counts1, kp_counts,
                       n, sel1, sel2, counts);
    return hipGetLastError();
}

hipError_t devmap_adaptive_sampling(DevMapScratch &S, size_t n, const AdaptiveBands &bands, int max_num_points, uint32_t *out_idx,
                                    size_t *out_count, hipStream_t stream) {
    *out_count = 0;
    if (n == 0) return hipSuccess;
    const unsigned grid = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(k_as_keys, dim3(grid), dim3(256), 0, stream, S.pts, S.stride, n, bands, S.keys, S.idx);
    DM_CHK(hipGetLastError());
    // stable sort on (band, z, y, x): indices stay ascending inside a voxel, so its first k positions are the k indices the
    // reference's loop pushes (sampling.h:80-85)
    DM_CHK(sort_pairs<uint64_t>(S.sort, S.keys, S.keys_alt, S.idx, S.idx_alt, n, 64, true, stream));
    hipLaunchKernelGGL(k_as_flags, dim3(grid), dim3(256), 0, stream, S.keys_alt, n, (uint32_t) bands.num_points_per_voxel, S.inserted);
    DM_CHK(hipGetLastError());
    DM_CHK(compact_flags(S.inserted, S.idx_alt, n, S.idx, S.sel_out, S.sel_count, stream));        // S.idx: free again after the sort
    DM_CHK(hipMemcpyAsync(S.h_count, S.sel_count, sizeof(int), hipMemcpyDeviceToHost, stream));
    DM_CHK(hipStreamSynchronize(stream));
    size_t count = (size_t) *S.h_count;
    if (max_num_points > 0 && count > (size_t) max_num_points + 1) count = (size_t) max_num_points + 1;   // `size() > max` (sampling.h:96-106)
    DM_CHK(hipMemcpyAsync(out_idx, S.sel_out, count * sizeof(uint32_t), hipMemcpyDefault, stream));   // host or device
    DM_CHK(hipStreamSynchronize(stream));
    *out_count = count;
    return hipSuccess;
}

hipError_t order_scratch_reserve(OrderScratch &S, size_t n) {
    if (n > S.cap) {
        order_scratch_free(S);
        const size_t cap = n + n / 4 + 1024;
        DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.keys), cap * sizeof(uint32_t)));
        DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.keys_alt), cap * sizeof(uint32_t)));
        DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.idx), cap * sizeof(uint32_t)));
        DM_CHK(hipMalloc(reinterpret_cast<void **>(&S.order), cap * sizeof(uint32_t)));
        DM_CHK(sort_scratch_reserve(S.sort, cap));
        S.cap = cap;
    }
    return hipSuccess;
}

hipError_t order_by_home_voxel(OrderScratch &S, const double *wx, const double *wy, const double *wz, size_t n, double resolution,
                               hipStream_t stream) {
    DM_CHK(order_scratch_reserve(S, n));
    hipLaunchKernelGGL(k_order_keys, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, wx, wy, wz, n, resolution, S.keys, S.idx);
    DM_CHK(hipGetLastError());
    return sort_pairs<uint32_t>(S.sort, S.keys, S.keys_alt, S.idx, S.order, n, 32, true, stream);
}

void order_scratch_free(OrderScratch &S) {
    if (S.keys) (void) hipFree(S.keys);
    if (S.keys_alt) (void) hipFree(S.keys_alt);
    if (S.idx) (void) hipFree(S.idx);
    if (S.order) (void) hipFree(S.order);
    sort_scratch_free(S.sort);
    S = OrderScratch{};
}

hipError_t devmap_test_sort(const uint64_t *keys_host, size_t n, int key_bits, int key_bytes, uint32_t *order_host, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    SortScratch S;
    void *k0 = nullptr, *k1 = nullptr;
    uint32_t *v0 = nullptr, *v1 = nullptr;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t r) {
        if (k0) (void) hipFree(k0);
        if (k1) (void) hipFree(k1);
        if (v0) (void) hipFree(v0);
        if (v1) (void) hipFree(v1);
        sort_scratch_free(S);
        return r;
    };
    if ((e = hipMalloc(&k0, n * 8)) != hipSuccess || (e = hipMalloc(&k1, n * 8)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void **>(&v0), n * 4)) != hipSuccess || (e = hipMalloc(reinterpret_cast<void **>(&v1), n * 4)) != hipSuccess)
        return done(e);
    if (key_bytes == 4) {
        std::vector<uint32_t> k32(n);
        for (size_t i = 0; i < n; ++i) k32[i] = (uint32_t) keys_host[i];
        if ((e = hipMemcpy(k0, k32.data(), n * 4, hipMemcpyHostToDevice)) != hipSuccess) return done(e);
        e = sort_pairs<uint32_t>(S, static_cast<uint32_t *>(k0), static_cast<uint32_t *>(k1), v0, v1, n, key_bits, true, stream);
    } else {
        if ((e = hipMemcpy(k0, keys_host, n * 8, hipMemcpyHostToDevice)) != hipSuccess) return done(e);
        e = sort_pairs<uint64_t>(S, static_cast<uint64_t *>(k0), static_cast<uint64_t *>(k1), v0, v1, n, key_bits, true, stream);
    }
    if (e != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
    e = hipMemcpy(order_host, v1, n * 4, hipMemcpyDeviceToHost);
    return done(e);
}

hipError_t devmap_test_compact(const uint8_t *flags_host, size_t n, uint32_t *out_host, size_t *count, hipStream_t stream) {
    *count = 0;
    if (n == 0) return hipSuccess;
    uint8_t *f = nullptr;
    uint32_t *cnt = nullptr, *out = nullptr;
    int *tot = nullptr;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t r) {
        if (f) (void) hipFree(f);
        if (cnt) (void) hipFree(cnt);
        if (out) (void) hipFree(out);
        if (tot) (void) hipFree(tot);
        return r;
    };
    if ((e = hipMalloc(reinterpret_cast<void **>(&f), n)) != hipSuccess || (e = hipMalloc(reinterpret_cast<void **>(&cnt), ((n + 255) / 256) * 4)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void **>(&out), n * 4)) != hipSuccess || (e = hipMalloc(reinterpret_cast<void **>(&tot), sizeof(int))) != hipSuccess)
        return done(e);
    if ((e = hipMemcpy(f, flags_host, n, hipMemcpyHostToDevice)) != hipSuccess) return done(e);
    if ((e = compact_flags(f, nullptr, n, cnt, out, tot, stream)) != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
    int t = 0;
    if ((e = hipMemcpy(&t, tot, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return done(e);
    *count = (size_t) t;
    e = hipMemcpy(out_host, out, (size_t) t * 4, hipMemcpyDeviceToHost);
    return done(e);
}

hipError_t devmap_level_remove_far(DevLevel &L, const double loc[3], double distance, hipStream_t stream) {
    hipLaunchKernelGGL(k_dm_remove_far, dim3((unsigned) std::min<uint64_t>((L.slots_cap + 255) / 256, 4096)), dim3(256), 0, stream, L.slots,
                       L.slots_cap, L.blocks, L.blk, L.free_list, L.counters, loc[0], loc[1], loc[2], distance, L.resolution, (const double *) nullptr,
                       (const int *) nullptr);
    DM_CHK(hipGetLastError());
    return read_counters(L, stream);
}

// the same without the counter read-back
hipError_t devmap_level_remove_far_value_enqueue(DevLevel &L, const double loc[3], double distance, hipStream_t stream) {
    hipLaunchKernelGGL(k_dm_remove_far, dim3((unsigned) std::min<uint64_t>((L.slots_cap + 255) / 256, 4096)), dim3(256), 0, stream, L.slots,
                       L.slots_cap, L.blocks, L.blk, L.free_list, L.counters, loc[0], loc[1], loc[2], distance, L.resolution, (const double *) nullptr,
                       (const int *) nullptr);
    return hipGetLastError();
}

// the same with the location read from device memory (3 doubles) and no counter read-back
hipError_t devmap_level_remove_far_enqueue(DevLevel &L, const double *loc_dev, double distance, hipStream_t stream, const int *failed_dev) {
    hipLaunchKernelGGL(k_dm_remove_far, dim3((unsigned) std::min<uint64_t>((L.slots_cap + 255) / 256, 4096)), dim3(256), 0, stream, L.slots,
                       L.slots_cap, L.blocks, L.blk, L.free_list, L.counters, 0.0, 0.0, 0.0, distance, L.resolution, loc_dev, failed_dev);
    return hipGetLastError();
}

hipError_t devmap_level_export(DevLevel &L, double *out_xyz, uint64_t cap_points, uint64_t *out_n, hipStream_t stream) {
    DM_CHK(read_counters(L, stream));
    if (out_n) *out_n = L.host.num_points;
    if (!out_xyz) return hipSuccess;
    std::vector<Slot> slots(L.slots_cap);
    std::vector<double> blocks((size_t) L.host.next_block * 3 * L.blk);
    DM_CHK(hipMemcpyAsync(slots.data(), L.slots, L.slots_cap * sizeof(Slot), hipMemcpyDeviceToHost, stream));
    if (!blocks.empty()) DM_CHK(hipMemcpyAsync(blocks.data(), L.blocks, blocks.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    DM_CHK(hipStreamSynchronize(stream));
    uint64_t k = 0;
    for (const Slot &s : slots) {
        if (s.key == KEY_EMPTY || s.key == KEY_TOMB) continue;
        const double *bx = &blocks[(size_t) s.block * 3 * L.blk];
        for (uint32_t j = 0; j < s.count; ++j, ++k)
            if (k < cap_points) { out_xyz[3 * k] = bx[3 * j]; out_xyz[3 * k + 1] = bx[3 * j + 1]; out_xyz[3 * k + 2] = bx[3 * j + 2]; }
    }
    return hipSuccess;
}

}  // namespace ctgn
