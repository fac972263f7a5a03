// ctgn_dense.hpp — the voxel-hash neighbour search of one GN iteration for scans whose keypoints crowd the voxels
// (BASELINE.json configs[1]: a 64-beam sweep puts ~45 keypoints into each 0.8 m home voxel), gfx950, wave64.
//
// Same results as k_accumulate_rows + the gather half of k_residual_reduce (ct_icp.cpp:753-769 with
// include/ct_icp/map.h:449-514 and include/SlamCore/experimental/neighborhood.h:236-240), organised the other way round:
//
//   k_accumulate_rows   16 lanes x 1 keypoint: the lanes of a row are CANDIDATES; every keypoint probes its own 27 / 125
//                       voxels and streams its own ~180 map points, although its scan neighbours walk the very same ones.
//   k_search_dense      64 lanes = 64 KEYPOINTS of consecutive sorted positions (ctgn_api.hip, order_keypoints). The lanes
//                       that share a home voxel form a run; the run's neighbourhood is probed ONCE by the wave, and its
//                       candidates are fed to all lanes from SCALAR registers (s_load through the scalar cache: the candidate
//                       is wave-uniform, the queries sit in the lanes), so one candidate costs ~15 vector instructions for
//                       up to 64 keypoints instead of for 4, with no per-lane address arithmetic and no vector-memory traffic.
//
// Selection of the k nearest without per-candidate heap work (the reference's bounded max-heap, map.h:494-500, keeps the k
// smallest distances; any exact k-selection under the order (d2, visit index) returns the same set):
//   pass 1  every in-radius candidate increments the lane's 64-bin histogram of d2 over [0, r2] (LDS, two 16-bit counters
//           per dword, fire-and-forget ds_add);
//   pivot   the first bin in which the running count reaches k; everything in lower bins is kept, the pivot bin holds the
//           candidates among which the last few are chosen;
//   pass 2  candidates of lower bins go to the lane's winner list, candidates of the pivot bin to its pivot list (LDS);
//   rank    the few pivot entries are ranked exactly by (d2, visit index) — d2 recomputed with the same arithmetic, so the
//           comparison is the one every other path makes;
//   sums    the lane walks its winners: mean / covariance sums (neighborhood.h:236-240) and the farthest kept neighbour
//           (= points[0] of the reference's farthest-first list, ct_icp.cpp:791), candidates read from an LDS copy of the
//           neighbourhood. The sums run in visit order, not in the reference's farthest-first order: another fixed order of
//           the same additions (rounding-level difference, inside the 1e-10 relative bar of the parity tests).
//   A lane whose pivot bin holds more than DN_PIV candidates (exact distance ties, lattice maps) is finished by the whole
//   wave with a plain exact selection (dense_fallback) — slow, rare, same result.
// Output per keypoint position: neighbour count, S = sum p, SS = sum p p^T (6), q = farthest kept — what the residual
// kernel needs; its 60 scattered gathers per keypoint are gone.
#pragma once

#include "ctgn_kernels.hpp"

namespace ctgn {

constexpr int DN_PC = 512;        // candidates of a run's neighbourhood mirrored in LDS (beyond: fetched from global memory)
constexpr int DN_BINS = 64;
constexpr int DN_LIST = 64;       // list entries per lane: winners grow from the front, pivot-bin entries from the back
constexpr int DN_PIV = 16;        // pivot-bin entries a lane ranks by itself

typedef const double __attribute__((address_space(4))) dn_cdouble;      // read through the scalar data cache

template <int NB>
struct DenseScratch {
    static constexpr int S = 2 * NB + 1, V = S * S * S;
    double cx[DN_PC], cy[DN_PC], cz[DN_PC];
    union {
        uint32_t hist32[DN_BINS * 32];           // [bin][lane >> 1], 16 bits per lane
        uint16_t list[DN_LIST * 64];             // [entry][lane]
    };
    uint32_t vox_off[V];                         // occupied voxels of the run, in sweep (= visit) order: byte offset of the block,
    uint16_t vox_cnt[V];                         // points in it,
    uint16_t vox_cbase[V + 1];                   // index of its first candidate
};

__device__ __forceinline__ int wave_scan_inclusive_i32(int v, int lane) {
    int s = row_scan_i32(v);
    const int t0 = __builtin_amdgcn_readlane(s, 15), t1 = __builtin_amdgcn_readlane(s, 31), t2 = __builtin_amdgcn_readlane(s, 47);
    const int row = lane >> 4;
    return s + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}
__device__ __forceinline__ int wave_max_i32(int v) {
    v = row_max_i32(v);
    return max_over_rows(v);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// xyz of candidate c of the current run: the LDS mirror, or (c >= DN_PC) the block storage via the occupied-voxel table
template <int NB>
__device__ __forceinline__ Vec3 dense_candidate(const DenseScratch<NB> &D, int c, int nocc, const char *pbase, uint32_t blk8) {
    if (c < DN_PC) return Vec3{D.cx[c], D.cy[c], D.cz[c]};
    int o = 0;
    while (o + 1 < nocc && (int) D.vox_cbase[o + 1] <= c) ++o;
    const char *p = pbase + D.vox_off[o] + 8u * (uint32_t) (c - (int) D.vox_cbase[o]);
    return Vec3{*reinterpret_cast<const double *>(p), *reinterpret_cast<const double *>(p + blk8),
                *reinterpret_cast<const double *>(p + 2 * blk8)};
}

// Whole-wave exact selection for the query held by lane L: the k smallest (d2, c) within the radius, one per step
// (every lane scans candidates lane, lane + 64, ...; a shuffle tree picks the minimum). The result goes straight to the
// hand-over arrays at lane L's position (no by-reference outputs: this is an out-of-line call).
template <int NB>
__device__ __noinline__ void dense_fallback(const DenseScratch<NB> *Dp, int L, int lane, int P, int nocc, const char *pbase, uint32_t blk8,
                                            int k, double r2thr, double qx, double qy, double qz, int pos, NbSums out) {
    const DenseScratch<NB> &D = *Dp;
    const double INF = __longlong_as_double(0x7ff0000000000000ll);
    const double ux = readlane_f64(qx, L), uy = readlane_f64(qy, L), uz = readlane_f64(qz, L);
    const int pos_L = __builtin_amdgcn_readlane(pos, L);
    double last_d2 = -1.0;
    int last_c = -1, nsel = 0;
    Vec3 S{0, 0, 0}, far{0, 0, 0};
    Sym3 SS{0, 0, 0, 0, 0, 0};
    for (int step = 0; step < k; ++step) {
        double bd2 = INF;
        int bc = 0x7fffffff;
        for (int c = lane; c < P; c += 64) {
            const Vec3 p = dense_candidate<NB>(D, c, nocc, pbase, blk8);
            const double d2 = sq_norm3(p.x - ux, p.y - uy, p.z - uz);
            const bool after = d2 > last_d2 || (d2 == last_d2 && c > last_c);
            if (d2 <= r2thr && after && (d2 < bd2 || (d2 == bd2 && c < bc))) { bd2 = d2; bc = c; }
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
            const double od2 = __shfl_xor(bd2, sh);
            const int oc = __shfl_xor(bc, sh);
            if (od2 < bd2 || (od2 == bd2 && oc < bc)) { bd2 = od2; bc = oc; }
        }
        if (!(bd2 < INF)) break;
        const Vec3 p = dense_candidate<NB>(D, bc, nocc, pbase, blk8);      // wave-uniform
        S = S + p;
        SS.xx += p.x * p.x; SS.xy += p.x * p.y; SS.xz += p.x * p.z; SS.yy += p.y * p.y; SS.yz += p.y * p.z; SS.zz += p.z * p.z;
        far = p;                                                          // ascending order: the last one taken is the farthest
        last_d2 = bd2; last_c = bc; ++nsel;
    }
    if (lane == 0) {
        out.cnt[pos_L] = (uint32_t) nsel;
        double *o = out.v + pos_L;
        const size_t s = out.stride;
        o[0] = S.x; o[s] = S.y; o[2 * s] = S.z;
        o[3 * s] = SS.xx; o[4 * s] = SS.xy; o[5 * s] = SS.xz; o[6 * s] = SS.yy; o[7 * s] = SS.yz; o[8 * s] = SS.zz;
        o[9 * s] = far.x; o[10 * s] = far.y; o[11 * s] = far.z;
    }
}

// One block = one wave = 64 consecutive positions. `rounds`-style tuning does not apply: a tile is 64 keypoints.
// PROF: shader clocks per phase, summed over waves into prof[0..9]: 0 phase A | 1 probes + table | 2 LDS mirror | 3 pass 1 |
// 4 pivot | 5 pass 2 | 6 pivot rank | 7 sums | 8 fallback | 9 hand-over; prof[10] = slowest wave, prof[11] = waves, prof[12] = runs
template <int NB, bool PROF = false>
__global__ __launch_bounds__(64) void k_search_dense(MapView map, KpView kp, const GnState *st, GnParams prm, NbSums out, int first_iter,
                                                     int ntiles, int write_all, unsigned long long *prof = nullptr, int ablate = 0) {
    constexpr int S = 2 * NB + 1, V = S * S * S;
    __shared__ DenseScratch<NB> D;
    if (st->done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0 && kp.clk_iter_start) *kp.clk_iter_start = wall_clock64();
    const int lane = threadIdx.x;
    const int k = prm.max_nb;
    const uint32_t blk = (uint32_t) map.blk, blk8 = blk * 8u, stride3 = 3u * blk8;
    const char *pbase = reinterpret_cast<const char *>(map.blocks);
    const double r2thr = map.r2thr;
    const double bin_scale = (double) DN_BINS / r2thr;
    unsigned long long pc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    if (PROF) tprev = __builtin_readcyclecounter();
#define DN_TICK(slot)                                                    \
    if (PROF) {                                                          \
        const unsigned long long now_ = __builtin_readcyclecounter();    \
        pc[slot] += now_ - tprev;                                        \
        tprev = now_;                                                    \
    }
    // contiguous tile ranges per XCD (workgroups are dealt round-robin to the 8 XCDs): each L2 then serves one eighth of the scan
    const int per = (ntiles + 7) >> 3;
    for (int b = blockIdx.x; b < per * 8; b += gridDim.x) {
        const int tile = (b & 7) * per + (b >> 3);
        if ((b >> 3) >= per || tile >= ntiles) continue;
        const int pos = tile * 64 + lane;
        const bool own = pos < kp.n;
        const int my_kp = own ? (kp.order ? (int) kp.order[pos] : pos) : -1;
        // ---------------- phase A: world point (re-transform with the current pose, ct_icp.cpp:964-966) and home voxel
        double qx = 0, qy = 0, qz = 0;
        int kx = INT_MIN, ky = 0, kz = 0;
        if (own) {
            Vec3 p;
            if (first_iter) {
                p = Vec3{kp.wx[my_kp], kp.wy[my_kp], kp.wz[my_kp]};
            } else {
                const Vec3 raw{kp.rx[my_kp], kp.ry[my_kp], kp.rz[my_kp]};
                const double alpha = alpha_timestamp(kp.t[my_kp], st->tbe[0], st->tbe[1]);
                p = ct_transform(st, alpha, raw);
                kp.wx[my_kp] = p.x; kp.wy[my_kp] = p.y; kp.wz[my_kp] = p.z;
            }
            qx = p.x; qy = p.y; qz = p.z;
            const int a = voxel_coord(p.x, map.resolution), bb = voxel_coord(p.y, map.resolution), c = voxel_coord(p.z, map.resolution);
            if (sweep_in_short_range(a, NB) && sweep_in_short_range(bb, NB) && sweep_in_short_range(c, NB)) { kx = a; ky = bb; kz = c; }
        }
        DN_TICK(0)
        int n_out = 0;
        bool written = false;                                   // finished (and written) by dense_fallback
        Vec3 S_out{0, 0, 0}, q_out{0, 0, 0};
        Sym3 SS_out{0, 0, 0, 0, 0, 0};

        unsigned long long todo = __ballot(own && kx != INT_MIN);
        while (todo) {
            // ---------------- the next run: all lanes of the tile that share the home voxel of the first open lane
            const int leader = __ffsll((long long) todo) - 1;
            const int hx = __builtin_amdgcn_readlane(kx, leader), hy = __builtin_amdgcn_readlane(ky, leader),
                      hz = __builtin_amdgcn_readlane(kz, leader);
            const bool member = own && kx == hx && ky == hy && kz == hz;
            const unsigned long long mm = __ballot(member);
            todo &= ~mm;

            // ---------------- probe the sweep voxels once for the run; occupied ones go to the table in sweep order
            int nocc = 0, P = 0;
#pragma unroll
            for (int v0 = 0; v0 < V; v0 += 64) {
                const int v = v0 + lane;
                uint32_t bc = 0u;
                if (v < V) bc = map_lookup(map, hx + v / (S * S) - NB, hy + (v / S) % S - NB, hz + v % S - NB);
                const int cnt = (int) (bc & 127u);
                const unsigned long long hb = __ballot(cnt > 0);
                const int incl = wave_scan_inclusive_i32(cnt, lane);
                if (cnt > 0) {
                    const int o = nocc + __popcll(hb & ((1ull << lane) - 1ull));
                    D.vox_off[o] = (bc >> 7) * stride3;
                    D.vox_cnt[o] = (uint16_t) cnt;
                    D.vox_cbase[o] = (uint16_t) (P + incl - cnt);
                }
                nocc += __popcll(hb);
                P += __builtin_amdgcn_readlane(incl, 63);
            }
            if (lane == 0) D.vox_cbase[nocc] = (uint16_t) P;
            // zero the histogram (it aliases the previous run's lists)
            {
                uint4 *h4 = reinterpret_cast<uint4 *>(D.hist32);
#pragma unroll
                for (int i = 0; i < DN_BINS * 32 / 4 / 64; ++i) h4[i * 64 + lane] = make_uint4(0u, 0u, 0u, 0u);
            }
            __builtin_amdgcn_wave_barrier();
            if (PROF) pc[12] += 1;
            DN_TICK(1)
            if (P == 0) continue;                                   // nothing around: these lanes keep n = 0

            // ---------------- mirror the neighbourhood in LDS (first DN_PC candidates), four voxels' loads in flight at a time
            for (int o0 = 0; o0 < nocc; o0 += 4) {
                double lx[4], ly[4], lz[4];
                int cc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int o = min(o0 + u, nocc - 1);
                    const int cnt = (o0 + u < nocc) ? (int) D.vox_cnt[o] : 0;
                    const bool ok = lane < cnt && (int) D.vox_cbase[o] + lane < DN_PC;
                    cc[u] = ok ? (int) D.vox_cbase[o] + lane : -1;
                    const char *p = pbase + D.vox_off[o] + (ok ? 8u * (uint32_t) lane : 0u);
                    lx[u] = *reinterpret_cast<const double *>(p);
                    ly[u] = *reinterpret_cast<const double *>(p + blk8);
                    lz[u] = *reinterpret_cast<const double *>(p + 2 * blk8);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (cc[u] >= 0) { D.cx[cc[u]] = lx[u]; D.cy[cc[u]] = ly[u]; D.cz[cc[u]] = lz[u]; }
            }
            __builtin_amdgcn_wave_barrier();
            DN_TICK(2)

            // the scalar-fed scan over all candidates of the run: body(c, x, y, z) with wave-uniform c, x, y, z
            auto scan = [&](auto &&body) {
                for (int o = 0; o < nocc; ++o) {
                    const uint32_t off = (uint32_t) __builtin_amdgcn_readfirstlane((int) D.vox_off[o]);
                    const int cnt = __builtin_amdgcn_readfirstlane((int) D.vox_cnt[o]);
                    const int cb = __builtin_amdgcn_readfirstlane((int) D.vox_cbase[o]);
                    dn_cdouble *bx = (dn_cdouble *) (unsigned long long) (pbase + off);
                    dn_cdouble *by = (dn_cdouble *) (unsigned long long) (pbase + off + blk8);
                    dn_cdouble *bz = (dn_cdouble *) (unsigned long long) (pbase + off + 2 * blk8);
                    int j = 0;
                    for (; j + 4 <= cnt; j += 4) {
                        const double x0 = bx[j], x1 = bx[j + 1], x2 = bx[j + 2], x3 = bx[j + 3];
                        const double y0 = by[j], y1 = by[j + 1], y2 = by[j + 2], y3 = by[j + 3];
                        const double z0 = bz[j], z1 = bz[j + 1], z2 = bz[j + 2], z3 = bz[j + 3];
                        body(cb + j, x0, y0, z0);
                        body(cb + j + 1, x1, y1, z1);
                        body(cb + j + 2, x2, y2, z2);
                        body(cb + j + 3, x3, y3, z3);
                    }
                    for (; j < cnt; ++j) body(cb + j, bx[j], by[j], bz[j]);
                }
            };

            int n_r = 0;                                            // candidates within the radius (map.h:491-493)
            int bstar = DN_BINS, need = 0, pivot_pop = 0;
            int nw = 0, np = 0;
            bool ovf = false;
            if (member) {
                // ---------------- pass 1: histogram of d2
                uint32_t *hcol = D.hist32 + (lane >> 1);
                const uint32_t hinc = 1u << (16 * (lane & 1));
                scan([&](int, double x, double y, double z) {
                    const double d2 = sq_norm3(x - qx, y - qy, z - qz);
                    if (d2 <= r2thr) {
                        const int bin = min(DN_BINS - 1, (int) (d2 * bin_scale));
                        if (!(ablate & 4)) atomicAdd(hcol + bin * 32, hinc);
                        ++n_r;
                    }
                });
                DN_TICK(3)
                // ---------------- pivot bin: the first one in which the running count reaches k
                if (n_r >= k && !(ablate & 6)) {
                    int cum = 0;
                    bstar = -1;
                    const int sh = 16 * (lane & 1);
                    for (int b0 = 0; b0 < DN_BINS; b0 += 16) {
                        uint32_t hv[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) hv[u] = hcol[(b0 + u) * 32];          // 16 independent LDS reads in flight
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int c = (int) ((hv[u] >> sh) & 0xffffu);
                            if (bstar < 0 && cum + c >= k) { bstar = b0 + u; need = k - cum; pivot_pop = c; }
                            cum += c;
                        }
                        if (!__any(bstar < 0)) break;
                    }
                    ovf = pivot_pop > DN_PIV;
                }
            }
            __builtin_amdgcn_wave_barrier();                        // every lane has read its histogram column: the lists may overwrite it
            DN_TICK(4)
            if (member && !ovf && !(ablate & 14)) {
                // ---------------- pass 2: winners (lower bins) to the front of the lane's list, pivot-bin entries to its back
                uint16_t *lcol = D.list + lane;
                scan([&](int c, double x, double y, double z) {
                    const double d2 = sq_norm3(x - qx, y - qy, z - qz);
                    const int bin = min(DN_BINS - 1, (int) (d2 * bin_scale));
                    const bool in = d2 <= r2thr;
                    const bool is_w = in && bin < bstar, is_p = in && bin == bstar;
                    // one store, no pointers to the counters (they stay in registers): winners at nw, pivot entries at the far end
                    if ((is_w || is_p) && !(ablate & 1)) lcol[(is_w ? nw : DN_LIST - 1 - np) * 64] = (uint16_t) c;
                    nw += is_w ? 1 : 0;
                    np += is_p ? 1 : 0;
                });
                if (ablate & 1) { nw = 0; np = 0; }
            }
            DN_TICK(5)
            // ---------------- exact rank inside the pivot bin: entry i is kept iff fewer than `need` entries precede it in (d2, c)
            {
                const int m = wave_max_i32((member && !ovf) ? np : 0);
                uint16_t *lcol = D.list + lane;
                for (int i = 0; i < m; ++i) {
                    const bool vi = member && !ovf && i < np;
                    const int ci = vi ? (int) lcol[(DN_LIST - 1 - i) * 64] : 0;
                    const Vec3 pi = dense_candidate<NB>(D, ci, nocc, pbase, blk8);
                    const double di = sq_norm3(pi.x - qx, pi.y - qy, pi.z - qz);
                    int rank = 0;
                    for (int j = 0; j < m; ++j) {
                        const bool vj = vi && j < np;
                        const int cj = vj ? (int) lcol[(DN_LIST - 1 - j) * 64] : 0;
                        const Vec3 pj = dense_candidate<NB>(D, cj, nocc, pbase, blk8);
                        const double dj = sq_norm3(pj.x - qx, pj.y - qy, pj.z - qz);
                        rank += (vj && (dj < di || (dj == di && cj < ci))) ? 1 : 0;
                    }
                    if (vi && rank < need) { lcol[nw * 64] = (uint16_t) ci; ++nw; }
                }
            }
            DN_TICK(6)
            // ---------------- sums over the winners, farthest kept neighbour
            {
                const int wmax = wave_max_i32((member && !ovf) ? nw : 0);
                const uint16_t *lcol = D.list + lane;
                Vec3 Sx{0, 0, 0}, far{0, 0, 0};
                Sym3 SS{0, 0, 0, 0, 0, 0};
                double fd2 = -1.0;
                int fc = -1;
                for (int w = 0; w < wmax; ++w) {
                    const bool vw = member && !ovf && w < nw;
                    const int c = vw ? (int) lcol[w * 64] : 0;
                    const Vec3 p = dense_candidate<NB>(D, c, nocc, pbase, blk8);
                    if (vw) {
                        Sx = Sx + p;
                        SS.xx += p.x * p.x; SS.xy += p.x * p.y; SS.xz += p.x * p.z;
                        SS.yy += p.y * p.y; SS.yz += p.y * p.z; SS.zz += p.z * p.z;
                        const double d2 = sq_norm3(p.x - qx, p.y - qy, p.z - qz);
                        if (d2 > fd2 || (d2 == fd2 && c > fc)) { fd2 = d2; fc = c; far = p; }
                    }
                }
                if (member && !ovf) { n_out = nw; S_out = Sx; SS_out = SS; q_out = far; }
            }
            DN_TICK(7)
            // ---------------- lanes whose pivot bin is crowded (exact ties): whole-wave exact selection, one query at a time
            unsigned long long fb = __ballot(member && ovf);
            while (fb) {
                const int L = __ffsll((long long) fb) - 1;
                fb &= fb - 1;
                dense_fallback<NB>(&D, L, lane, P, nocc, pbase, blk8, k, r2thr, qx, qy, qz, pos, out);
            }
            written = written || (member && ovf);
            __builtin_amdgcn_wave_barrier();
            DN_TICK(8)
        }
        // ---------------- hand-over, by position: count always; sums only where the gates can keep the keypoint (ct_icp.cpp:769)
        if (own && !written) {
            out.cnt[pos] = (uint32_t) n_out;
            if ((n_out >= prm.min_nb && n_out >= 5) || write_all) {
                double *o = out.v + pos;
                const size_t s = out.stride;
                o[0] = S_out.x; o[s] = S_out.y; o[2 * s] = S_out.z;
                o[3 * s] = SS_out.xx; o[4 * s] = SS_out.xy; o[5 * s] = SS_out.xz; o[6 * s] = SS_out.yy; o[7 * s] = SS_out.yz; o[8 * s] = SS_out.zz;
                o[9 * s] = q_out.x; o[10 * s] = q_out.y; o[11 * s] = q_out.z;
            }
        }
        DN_TICK(9)
    }
    if (PROF && lane == 0) {
        unsigned long long tot_ = 0;
        for (int q = 0; q < 10; ++q) { atomicAdd(&prof[q], q == 9 ? pc[12] : pc[q]); tot_ += pc[q]; }      // slot 9 reports the runs
        atomicMax(&prof[10], tot_);
        atomicAdd(&prof[11], 1ull);
        atomicAdd(&prof[12], pc[12]);
    }
#undef DN_TICK
}

}  // namespace ctgn
