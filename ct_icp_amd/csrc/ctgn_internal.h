/* ctgn_internal.h — measurement and test hooks of libctgn.so that are NOT part of the drop-in contract (include/ctgn.h).
 *
 * The reference-side binding (integration/) sees include/ctgn.h only. What is declared here is exported by the same
 * shared library for this repository's own tests, profiling scripts and bench.py: ablation masks of the row kernel, the instrumented
 * instantiation's phase clocks and traffic counters, the per-wave timeline, and direct entry points to the library's sort / compaction
 * kernels. Nothing here is stable ABI. */
#ifndef CTGN_INTERNAL_H
#define CTGN_INTERNAL_H

#include "../../include/ctgn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement hook: skip phases of the row kernel (bit 0 candidate streaming, 1 final selection, 2 hand-over records,
 * 3 normal/residual/Jacobian, 4 hash probes, 5 shared-home-voxel path (variant 5), 6 radius cull of the probes off, 7 no list
 * appends, 8 no carried-over bound and no pools, 9 no slab masks, 10 no rounds at all = phase A only, 11 no neighbour pools,
 * 12-15 spare pool members instead of 8 (1..12)). Results are INVALID while a mask is set — except bits 5, 6, 8, 9 and 11-15, which only
 * switch exact optimisations off or tune them — and a starved solve keeps launching the search so that it can be timed (ablation
 * profiling, DESIGN.md sections 13 and 17, scripts/ablate2.sh, scripts/iter_times.py). */
ctgn_status ctgn_set_ablation(ctgn_handle h, int32_t mask);
/* How the GN route solves a neighbourhood's 3 x 3 covariance (ctgn_kernels.hpp, residual_jacobian): 0 = the library default (2), 1 = exact
 * (Eigen's JacobiSVD restated, correctly rounded: normals and a2D bit-identical to the reference build's; +11 % per iteration on a
 * 132 k-keypoint sweep), 2 = hybrid (fast solver, the exact one only where a gate decision or a near-degenerate neighbourhood could depend
 * on the difference: identical gate decisions and n_used, normals to ~1e-12), 3 = fast only. Takes effect at the next ctgn_gn_begin /
 * solve. Bits 16-17 of the ablation mask override it; bit 18 toggles the first-search cull of the 27-voxel sweep's second probe batch,
 * bit 19 switches the split pool-check launches off and bit 25 forces them on below their size threshold, bit 22 switches the wave-shared
 * probes of the 125-voxel sweep off, bit 24 the guessed first-search bound, bit 26 the 125-voxel sweep's early stop after the probe batches
 * some row can reach, bit 27 the reuse of the wave's probe table across rounds of one home voxel, bit 29 the pools of searches whose
 * carried-over bound lies beyond the radius (A/B hooks; these bits leave results valid). */
ctgn_status ctgn_set_normals(ctgn_handle h, int32_t mode);
/* The guessed bound of a first search (ctgn_api.hip, launch_accumulate; rows_tiles pass 1): factor < 0 = automatic (1.25 x the radius k
 * neighbours fill on a surface at the searched level's points per voxel, used only when its square is below 0.8 of the squared search radius), 0 = off,
 * > 0 = that factor, used whenever the guess is inside the radius (tests force small factors so that most guesses fail and the second
 * pass runs). Results do not depend on it. Bit 24 of the ablation mask also switches it off. */
ctgn_status ctgn_set_search_guess(ctgn_handle h, double factor);
/* Shader-clock cycles summed over all waves of the variant-3 launches since the last reset, per phase:
 * 0 transform+voxel, 1 hash probes, 2 candidate streaming, 3 in-stream prunes, 4 final selection,
 * 5 covariance sums, 6 normal+residual+Jacobian, 7 u u^T accumulation; 8 = rounds that took the shared-home-voxel
 * fast path, 9 = rounds (counts, per wave), 10 = clocks of the slowest wave (max), 11 = waves. */
ctgn_status ctgn_phase_cycles(ctgn_handle h, uint64_t out[12], int32_t reset);
/* What the instrumented row kernel (variant 3) actually requested from the memory system since the last reset: out[0] = hash probes
 * issued (sweep voxels NOT culled against the radius / the k-th best distance; the shared-home-voxel path probes its 27 voxels once
 * per wave), out[1] = map points streamed. bench.py prices the kernel's algorithmic bytes with these instead of
 * charging every point of all 27 / 125 sweep voxels. Measurement hook. */
ctgn_status ctgn_traffic_counters(ctgn_handle h, uint64_t out[2], int32_t reset);
/* Per-wave timeline of the last variant-3 launch: 4 words per wave slot (start clock, end clock, fast-path rounds,
 * rounds); slots of waves that did not run keep their previous content (zero initially). */
ctgn_status ctgn_wave_timeline(ctgn_handle h, uint64_t *out, size_t max_waves, size_t *n_waves);

/* Moved here from include/ctgn.h in round 5 (measurement / A-B hooks, not part of the drop-in contract): */
/* Counting pass for the roofline: V = voxels probed per keypoint, total map points inside them, summed
 * over the resident keypoints at their current world positions. */
ctgn_status ctgn_count_traffic(ctgn_handle h, uint64_t *voxels_probed, uint64_t *voxels_hit,
                               uint64_t *points_scanned);
/* The same, split by what bounded the search: [0] = the first search of a solve (radius only: nothing carried over), [1] = every later
 * one (bounded by the previous search's k-th neighbour distance + the keypoint's displacement). */
ctgn_status ctgn_kernel_timing_split(ctgn_handle h, double avg_ms[2], int32_t launches[2], int32_t reset);
/* Select the accumulate kernel: 0 = 16-lanes-per-keypoint + histogram-assisted selection (default),
 * 1 = lane-per-keypoint cross-check kernel, 2 = 16-lanes-per-keypoint with plain rank selection,
 * 3 = variant 0 instrumented with per-phase shader clocks (read through ct_icp_amd/csrc/ctgn_internal.h), 4 = variant 0 compiled for
 * 4 waves per SIMD instead of 3 (A/B hook), 5 = variant 0 with the shared-home-voxel path of the 27-voxel sweep compiled in
 * (the four keypoints of a round probe and stream one flattened neighbourhood; the default until round 2, now slower than the
 * bounded generic path — A/B hook). Same results for every variant. Test / measurement hook. */
ctgn_status ctgn_set_variant(ctgn_handle h, int32_t variant);

/* Measured streaming bandwidth of the device's HBM (the roofline's peak verified on the box): float4 copy and triad over
 * `bytes`-sized arrays (>= 1 MiB; three of them are allocated for the call), `reps` timed launches each. out_gbs[0] = copy, [1] = triad,
 * [2] = the runtime's own device-to-device hipMemcpyAsync; GB/s of bytes read + written. Measurement hook (bench.py: roofline.peak_measured). */
ctgn_status ctgn_measure_hbm(ctgn_handle h, uint64_t bytes, int32_t reps, double out_gbs[3]);

/* The A/B switches of the measurement sessions, one table (ctgn_api.hip, struct Tuning: host_threads, order, pool_min, res_small, res_grid_cap,
 * guess_factor, guess_maxfrac, split, xcd_split, fuse_small, persistent, persist_times, frame_timing, frame_no_direct, frame_defer_update,
 * stop_poll, tile_chunk, stage_lds, state_init_fused, xcd_reduce, robust_fuse). Process-wide and NOT synchronised: for a script or a test that owns the process, between
 * calls. None of them changes a neighbour set, a gate decision or a per-keypoint quantity; order, xcd_reduce, fuse_small and robust_fuse
 * select another fixed order of the packed sums (poses move in their last bits; see the table's comment). CTGN_ERR_UNSUPPORTED: host_threads
 * after the helper pool was sized by the first scan-sized frame call. A session script that cannot call into the library sets
 * CTGN_TUNING="key=value,key=value" instead (read once). */
ctgn_status ctgn_set_tuning(const char *key, double value);

/* Has the size-dependent reduction path run (tests assert that the path they mean to cover did)? out[0] = residual launches of this handle
 * whose block records got per-XCD pre-sums (XcdReduce, ctgn_kernels.hpp), out[1] = 1 if the last solve launch summed the group records
 * rather than the block records (the placement check held). Drains the stream. */
ctgn_status ctgn_path_counters(ctgn_handle h, uint64_t out[2]);

/* Measurement hook: the carried per-keypoint search state (KpView::kth pairs, record count words), in working order. */
ctgn_status ctgn_debug_pool_state(ctgn_handle h, float *kth_out, uint32_t *cnt_out, size_t n);

/* Host-to-device bytes the last ctgn_set_keypoints_sharded call moved on this rank (host views): 56 B per keypoint of the scan with one rank,
 * 24 B per keypoint of the scan (world points, for the order every rank must agree on) + 56 B per keypoint of the rank's chunk otherwise. */
ctgn_status ctgn_last_upload_bytes(ctgn_handle h, uint64_t *bytes);

/* Test hooks for the library's own sort / compaction kernels (ctgn_sort.hpp; they replace a vendor sort library on the frame path: the
 * map-update batch, the adaptive sampler, the home-voxel ordering): order_out[j] = index of the j-th smallest key under a STABLE sort on
 * the low key_bits bits (key_bytes 4: the keys are narrowed to 32 bits first); out_indices = ascending indices whose flag is non-zero.
 * Host arrays in and out. */
ctgn_status ctgn_test_sort_pairs(ctgn_handle h, const uint64_t *keys, size_t n, int32_t key_bits, int32_t key_bytes, uint32_t *order_out);
ctgn_status ctgn_test_compact(ctgn_handle h, const uint8_t *flags, size_t n, uint32_t *out_indices, size_t *out_count);

/* What a sharded iteration costs before it does any work (round 6; bench.py puts both into the N > 1 line so that a scaling record explains
 * itself): out_us[0] = one ncclAllReduce(sum) of the 96-double packed system on the handle's stream, mean of `reps` back-to-back calls
 * (the collective of ctgn_solve_sharded, bare); out_us[1] = one iteration of that loop when its four kernels have nothing to do and return
 * at once (search, residual, reduce, all-reduce, solve: the launch and dependency overhead of the chain, all-reduce included). Collective:
 * every rank of the communicator must call it. Needs ctgn_dist_init. */
ctgn_status ctgn_dist_overheads(ctgn_handle h, int32_t reps, double out_us[2]);
#ifdef __cplusplus
}
#endif
#endif /* CTGN_INTERNAL_H */
