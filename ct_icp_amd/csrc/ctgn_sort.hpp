// ctgn_sort.hpp — the two data-movement primitives the frame path needs, written for gfx950 wave64 (no library underneath):
//   sort_pairs    stable least-significant-digit radix sort of (key, 32-bit value) pairs, 8-bit digits, keys of 32 or 64 bits;
//                 deterministic (integer ranks only, no floating point, no scatter atomics): every run and every rank of a sharded
//                 job produces the same order. Digits that are the same in every key are detected once (OR / AND of the keys) and
//                 their passes skipped — a frame's voxel keys vary in ~4 of their 8 bytes.
//                   n <= SORT_SMALL_MAX  ONE launch: a single 1024-thread block runs every pass (the map-update batch of a frame,
//                                        ~8 k keys: insert rule of include/ct_icp/map.h:261-293 needs the batch grouped by voxel in
//                                        original order);
//                   larger n             per executed pass: per-wave digit histograms -> one-block exclusive scan -> stable scatter.
//   compact_flags ordered stream compaction (indices or values whose flag is set, ascending), two launches, no scan kernel: every
//                 block adds up the per-block counts in front of it.
// Ranking inside a wave: the lanes that hold the same digit find each other with eight ballots (one per digit bit); a lane's rank
// among them is a popcount of the peer mask below it; the lowest peer bumps the wave's own LDS counter. Waves own contiguous element
// ranges, so (wave, round, lane) order IS element order and the sort is stable without any cross-wave traffic inside a pass.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "ctgn_devmap.hpp"      // SortScratch

namespace ctgn {

constexpr int SORT_SMALL_MAX = 16384;
constexpr int SORT_SMALL_THREADS = 1024;
constexpr int SORT_MAX_COLS = 1024;            // per-wave tiles of the large path (columns of the histogram matrix): the tile grows with n

inline hipError_t sort_scratch_reserve(SortScratch &S, size_t n) {
    (void) n;
    if (S.hist && S.bits) return hipSuccess;     // fixed size: the tile grows with n so that the columns stay <= SORT_MAX_COLS
    hipError_t e = hipSuccess;
    if (!S.hist) e = hipMalloc(reinterpret_cast<void **>(&S.hist), (size_t) 256 * SORT_MAX_COLS * sizeof(uint32_t));
    if (e == hipSuccess && !S.bits) e = hipMalloc(reinterpret_cast<void **>(&S.bits), (2 + 128) * sizeof(unsigned long long));     // OR | AND | 256 digit totals
    if (e != hipSuccess) {                       // all or nothing: a half-reserved scratch must not look reserved to the next call
        if (S.hist) (void) hipFree(S.hist);
        if (S.bits) (void) hipFree(S.bits);
        S.hist = nullptr; S.bits = nullptr;
    }
    return e;
}

inline void sort_scratch_free(SortScratch &S) {
    if (S.hist) (void) hipFree(S.hist);
    if (S.bits) (void) hipFree(S.bits);
    S = SortScratch{};
}

// lanes of the wave whose (valid) element has the same 8-bit digit as this lane's
__device__ __forceinline__ unsigned long long digit_peers(uint32_t d, bool valid) {
    unsigned long long peers = ballot64(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = ballot64(bit);
        peers &= bit ? m : ~m;
    }
    return valid ? peers : 0ull;
}

// does pass p (digit p) move anything, and in which buffer does it find its input (0 = primary, 1 = alternate)?
__device__ __forceinline__ bool pass_runs(unsigned long long varying, int p) { return ((varying >> (8 * p)) & 0xffull) != 0ull; }
__device__ __forceinline__ int buffer_before(unsigned long long varying, int p) {
    int c = 0;
    for (int q = 0; q < p; ++q) c += pass_runs(varying, q) ? 1 : 0;
    return c & 1;
}

// ------------------------------------------------------------------------------------------------ n <= SORT_SMALL_MAX: one block
// Result always ends in (keys_alt, vals_alt). iota_in: the values are 0 .. n-1 and `vals` holds nothing yet (it is still used as the
// ping-pong partner of vals_alt).
template <typename K>
__global__ __launch_bounds__(SORT_SMALL_THREADS) void k_sort_small(K *keys, K *keys_alt, uint32_t *vals, uint32_t *vals_alt, int n, int passes,
                                                                   int iota_in) {
    constexpr int NW = SORT_SMALL_THREADS / 64;
    __shared__ uint32_t s_cnt[NW][256];
    __shared__ uint32_t s_tot[256], s_wsum[4];
    __shared__ unsigned long long s_or[NW], s_and[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    // which digits vary at all
    unsigned long long vo = 0ull, va = ~0ull;
    for (int i = tid; i < n; i += SORT_SMALL_THREADS) { const unsigned long long k = (unsigned long long) keys[i]; vo |= k; va &= k; }
    for (int d = 32; d >= 1; d >>= 1) { vo |= __shfl_xor(vo, d); va &= __shfl_xor(va, d); }
    if (lane == 0) { s_or[wave] = vo; s_and[wave] = va; }
    __syncthreads();
    vo = 0ull; va = ~0ull;
    for (int w = 0; w < NW; ++w) { vo |= s_or[w]; va &= s_and[w]; }
    const unsigned long long varying = vo & ~va;
    const int chunk = (((n + NW - 1) / NW) + 63) & ~63;               // contiguous range of a wave, whole rounds
    const int e0 = wave * chunk, e1 = min(n, e0 + chunk);
    K *src_k = keys, *dst_k = keys_alt;
    uint32_t *src_v = vals, *dst_v = vals_alt;
    bool iota = iota_in != 0;                                          // values are the element indices until the first pass has run
    for (int p = 0; p < passes; ++p) {
        if (!pass_runs(varying, p)) continue;
        const int shift = 8 * p;
        for (int i = lane; i < 256; i += 64) s_cnt[wave][i] = 0u;
        // count (wave-private row; the LDS operations of a wave execute in order)
        for (int e = e0 + lane; e - lane < e1; e += 64) {
            const bool valid = e < e1;
            const uint32_t d = valid ? (uint32_t) ((src_k[e] >> shift) & 0xff) : 0u;
            const unsigned long long peers = digit_peers(d, valid);
            if (valid && (peers & below) == 0ull) s_cnt[wave][d] += (uint32_t) __popcll(peers);
        }
        __syncthreads();
        // exclusive offsets: over the waves per digit, then over the digits
        uint32_t tot = 0u;
        if (tid < 256) {
            for (int w = 0; w < NW; ++w) { const uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = tot; tot += c; }
            uint32_t inc = tot;
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
            if (lane == 63) s_wsum[wave] = inc;
            s_tot[tid] = inc - tot;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t base = s_tot[tid];
            for (int w = 0; w < wave; ++w) base += s_wsum[w];
            for (int w = 0; w < NW; ++w) s_cnt[w][tid] += base;
        }
        __syncthreads();
        // stable scatter
        for (int e = e0 + lane; e - lane < e1; e += 64) {
            const bool valid = e < e1;
            const K key = valid ? src_k[e] : (K) 0;
            const uint32_t val = !valid ? 0u : iota ? (uint32_t) e : src_v[e];
            const uint32_t d = (uint32_t) ((key >> shift) & 0xff);
            const unsigned long long peers = digit_peers(d, valid);
            if (valid) {
                const uint32_t pos = s_cnt[wave][d] + (uint32_t) __popcll(peers & below);
                dst_k[pos] = key;
                dst_v[pos] = val;
            }
            if (valid && (peers & below) == 0ull) s_cnt[wave][d] += (uint32_t) __popcll(peers);
        }
        __threadfence_block();
        __syncthreads();
        K *tk = src_k; src_k = dst_k; dst_k = tk;
        uint32_t *tv = src_v; src_v = dst_v; dst_v = tv;
        iota = false;
    }
    // the result is in src_*: move it to the alternate buffers if it is not there
    if (src_k != keys_alt) {
        for (int i = tid; i < n; i += SORT_SMALL_THREADS) {
            keys_alt[i] = src_k[i];
            vals_alt[i] = iota ? (uint32_t) i : src_v[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------ larger n: three kernels per pass
template <typename K>
__global__ __launch_bounds__(256) void k_sort_bits(const K *keys, size_t n, unsigned long long *bits) {
    __shared__ unsigned long long s_or[4], s_and[4];
    unsigned long long vo = 0ull, va = ~0ull;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        const unsigned long long k = (unsigned long long) keys[i];
        vo |= k; va &= k;
    }
    for (int d = 32; d >= 1; d >>= 1) { vo |= __shfl_xor(vo, d); va &= __shfl_xor(va, d); }
    if ((threadIdx.x & 63) == 0) { s_or[threadIdx.x >> 6] = vo; s_and[threadIdx.x >> 6] = va; }
    __syncthreads();
    // one pair of atomics per BLOCK (round 3 issued one per wave from 1 024 blocks: 8 k atomics on two addresses, 96 us at 0.9 M keys)
    if (threadIdx.x == 0) {
        atomicOr(&bits[0], s_or[0] | s_or[1] | s_or[2] | s_or[3]);
        atomicAnd(&bits[1], s_and[0] & s_and[1] & s_and[2] & s_and[3]);
    }
}

// wave (blockIdx * 4 + wave) owns elements [col * tile, (col + 1) * tile): its digit counts go to hist[digit][col]
template <typename K>
__global__ __launch_bounds__(256) void k_sort_hist(const K *k0, const K *k1, size_t n, int p, int tile, int cols, const unsigned long long *bits,
                                                   uint32_t *hist) {
    __shared__ uint32_t s_cnt[4][256];
    const unsigned long long varying = bits[0] & ~bits[1];
    if (!pass_runs(varying, p)) return;
    const K *src = buffer_before(varying, p) ? k1 : k0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = blockIdx.x * 4 + wave, shift = 8 * p;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int i = lane; i < 256; i += 64) s_cnt[wave][i] = 0u;
    if (col < cols) {
        const size_t e0 = (size_t) col * tile, e1 = std::min(n, e0 + (size_t) tile);
        for (size_t e = e0 + lane; e - lane < e1; e += 64) {
            const bool valid = e < e1;
            const uint32_t d = valid ? (uint32_t) ((src[e] >> shift) & 0xff) : 0u;
            const unsigned long long peers = digit_peers(d, valid);
            if (valid && (peers & below) == 0ull) s_cnt[wave][d] += (uint32_t) __popcll(peers);
        }
        for (int i = lane; i < 256; i += 64) hist[(size_t) i * cols + col] = s_cnt[wave][i];
    }
}

// Exclusive scan of hist[256][cols] (digit-major), in two levels: block d scans ROW d (all columns of digit d; cols <= 1024: one uint4 per
// thread) in place and leaves the row's total in totals[d]; the scatter kernel adds the exclusive scan of the 256 totals itself. (Round 3
// scanned all 256 x cols counters with one block, strip after strip: 35 us per pass at 0.9 M keys.)
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t *hist, int cols, int p, const unsigned long long *bits, uint32_t *totals) {
    __shared__ uint32_t s_wave[4];
    const unsigned long long varying = bits[0] & ~bits[1];
    if (!pass_runs(varying, p)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *row = hist + (size_t) blockIdx.x * cols;
    const int i = 4 * tid;
    uint32_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i + q < cols ? row[i + q] : 0u;
    const uint32_t own = v[0] + v[1] + v[2] + v[3];
    uint32_t inc = own;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = inc - own;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    uint32_t run = before;
#pragma unroll
    for (int q = 0; q < 4; ++q) { if (i + q < cols) row[i + q] = run; run += v[q]; }
    if (tid == 255) totals[blockIdx.x] = before + own;
}

template <typename K>
__global__ __launch_bounds__(256) void k_sort_scatter(K *k0, K *k1, uint32_t *v0, uint32_t *v1, size_t n, int p, int tile, int cols,
                                                      const unsigned long long *bits, const uint32_t *hist, int iota_first, const uint32_t *totals) {
    __shared__ uint32_t s_cnt[4][256];
    __shared__ uint32_t s_base[256], s_wsum[4];
    const unsigned long long varying = bits[0] & ~bits[1];
    if (!pass_runs(varying, p)) return;
    {   // where digit d's run starts: exclusive scan of the 256 row totals (thread t = digit t)
        const int t = threadIdx.x, l = t & 63, w = t >> 6;
        const uint32_t tot = totals[t];
        uint32_t inc = tot;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (l >= d) inc += o; }
        if (l == 63) s_wsum[w] = inc;
        __syncthreads();
        uint32_t base = inc - tot;
        for (int q = 0; q < w; ++q) base += s_wsum[q];
        s_base[t] = base;
        __syncthreads();
    }
    const int cur = buffer_before(varying, p);
    const bool first = iota_first && (varying & ((1ull << (8 * p)) - 1ull)) == 0ull;      // no earlier pass ran: values are still 0 .. n-1
    const K *src_k = cur ? k1 : k0;
    K *dst_k = cur ? k0 : k1;
    const uint32_t *src_v = cur ? v1 : v0;
    uint32_t *dst_v = cur ? v0 : v1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = blockIdx.x * 4 + wave, shift = 8 * p;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (col >= cols) return;
    for (int i = lane; i < 256; i += 64) s_cnt[wave][i] = hist[(size_t) i * cols + col] + s_base[i];
    const size_t e0 = (size_t) col * tile, e1 = std::min(n, e0 + (size_t) tile);
    for (size_t e = e0 + lane; e - lane < e1; e += 64) {
        const bool valid = e < e1;
        const K key = valid ? src_k[e] : (K) 0;
        const uint32_t val = !valid ? 0u : first ? (uint32_t) e : src_v[e];
        const uint32_t d = (uint32_t) ((key >> shift) & 0xff);
        const unsigned long long peers = digit_peers(d, valid);
        if (valid) {
            const uint32_t pos = s_cnt[wave][d] + (uint32_t) __popcll(peers & below);
            dst_k[pos] = key;
            dst_v[pos] = val;
        }
        if (valid && (peers & below) == 0ull) s_cnt[wave][d] += (uint32_t) __popcll(peers);
    }
}

// after the last pass: the result must sit in (k1, v1)
template <typename K>
__global__ __launch_bounds__(256) void k_sort_finish(const K *k0, K *k1, const uint32_t *v0, uint32_t *v1, size_t n, int passes,
                                                     const unsigned long long *bits, int iota_first) {
    const unsigned long long varying = bits[0] & ~bits[1];
    if (buffer_before(varying, passes) == 1) return;
    const bool untouched = iota_first && (passes >= 8 ? varying : (varying & ((1ull << (8 * passes)) - 1ull))) == 0ull;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        k1[i] = k0[i];
        v1[i] = untouched ? (uint32_t) i : v0[i];
    }
}

// Stable sort of n pairs by the low `key_bits` bits of the key. Input in (keys, vals) — vals == nullptr or iota_vals: the values are
// 0 .. n-1 and need not be written beforehand — result in (keys_alt, vals_alt); all four buffers are clobbered.
template <typename K>
inline hipError_t sort_pairs(SortScratch &S, K *keys, K *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t n, int key_bits, bool iota_vals,
                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int passes = std::min<int>((key_bits + 7) / 8, (int) sizeof(K));
    if (n <= (size_t) SORT_SMALL_MAX) {
        hipLaunchKernelGGL(k_sort_small<K>, dim3(1), dim3(SORT_SMALL_THREADS), 0, stream, keys, keys_alt, vals, vals_alt, (int) n, passes,
                           iota_vals ? 1 : 0);
        return hipGetLastError();
    }
    hipError_t e = sort_scratch_reserve(S, n);
    if (e != hipSuccess) return e;
    int tile = 1024;
    while (((n + (size_t) tile - 1) / (size_t) tile) > (size_t) SORT_MAX_COLS) tile *= 2;
    const int cols = (int) ((n + (size_t) tile - 1) / (size_t) tile);
    e = hipMemsetAsync(S.bits, 0x00, sizeof(unsigned long long), stream);                   // OR accumulator
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(S.bits + 1, 0xFF, sizeof(unsigned long long), stream);               // AND accumulator
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sort_bits<K>, dim3((unsigned) std::min<size_t>((n + 255) / 256, 256)), dim3(256), 0, stream, (const K *) keys, n, S.bits);
    uint32_t *totals = reinterpret_cast<uint32_t *>(S.bits + 2);
    const unsigned grid = (unsigned) ((cols + 3) / 4);
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(k_sort_hist<K>, dim3(grid), dim3(256), 0, stream, (const K *) keys, (const K *) keys_alt, n, p, tile, cols,
                           (const unsigned long long *) S.bits, S.hist);
        hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, stream, S.hist, cols, p, (const unsigned long long *) S.bits, totals);
        hipLaunchKernelGGL(k_sort_scatter<K>, dim3(grid), dim3(256), 0, stream, keys, keys_alt, vals, vals_alt, n, p, tile, cols,
                           (const unsigned long long *) S.bits, (const uint32_t *) S.hist, iota_vals ? 1 : 0, (const uint32_t *) totals);
    }
    hipLaunchKernelGGL(k_sort_finish<K>, dim3((unsigned) std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, stream, (const K *) keys, keys_alt,
                       (const uint32_t *) vals, vals_alt, n, passes, (const unsigned long long *) S.bits, iota_vals ? 1 : 0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ ordered compaction
__global__ __launch_bounds__(256) void k_flag_counts(const uint8_t *flags, size_t n, uint32_t *block_counts) {
    __shared__ uint32_t s_cnt[4];
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    const bool f = i < n && flags[i];
    const uint32_t c = (uint32_t) __popcll(ballot64(f));
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// out receives, in ascending position order, values[i] (or i when values == nullptr) of every position whose flag is set; *total the count
__global__ __launch_bounds__(256) void k_compact1(const uint8_t *flags, const uint32_t *values, const uint32_t *counts, size_t n, uint32_t *out,
                                                  int *total) {
    __shared__ uint32_t s_red[4], s_wave[4];
    uint32_t pa = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256) pa += counts[b];
    for (int d = 32; d >= 1; d >>= 1) pa += __shfl_xor(pa, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = pa;
    const size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    const bool f = i < n && flags[i];
    const unsigned long long m = ballot64(f);
    if (lane == 0) s_wave[wave] = (uint32_t) __popcll(m);
    __syncthreads();
    uint32_t base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (f) out[base + (uint32_t) __popcll(m & below)] = values ? values[i] : (uint32_t) i;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *total = (int) (base + (uint32_t) __popcll(m & below) + (f ? 1u : 0u));
}

// counts: scratch of >= ceil(n / 256) words
inline hipError_t compact_flags(const uint8_t *flags, const uint32_t *values, size_t n, uint32_t *counts, uint32_t *out, int *total, hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(total, 0, sizeof(int), stream);
    const unsigned grid = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(k_flag_counts, dim3(grid), dim3(256), 0, stream, flags, n, counts);
    hipLaunchKernelGGL(k_compact1, dim3(grid), dim3(256), 0, stream, flags, values, (const uint32_t *) counts, n, out, total);
    return hipGetLastError();
}

}  // namespace ctgn
