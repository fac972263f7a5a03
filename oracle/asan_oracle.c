/* asan_oracle.c — drives the CPU oracle (ctgn_oracle.c, ctgn_oracle_robust.c) through its whole surface on a small seeded scene under
 * -fsanitize=address,undefined (`make -C oracle asan`; tests/test_sanitizers.py runs it). Test infrastructure: SURVEY.md section 5's
 * sanitizer row applied to the checker itself. Exits 0 when no sanitizer fired and the results are sane. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ctgn_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double) (rng_state >> 11) / 9007199254740992.0;
}
static double nrand(void) { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

/* a room: floor, two walls, noise 1 cm */
static void room_point(double *p) {
    const int s = (int) (urand() * 3.0);
    const double a = urand() * 12.0 - 6.0, b = urand() * 12.0 - 6.0;
    if (s == 0) { p[0] = a; p[1] = b; p[2] = 0.0; }
    else if (s == 1) { p[0] = a; p[1] = 6.0; p[2] = 0.25 * (b + 6.0); }
    else { p[0] = -6.0; p[1] = a; p[2] = 0.25 * (b + 6.0); }
    for (int c = 0; c < 3; ++c) p[c] += 0.01 * nrand();
}

int main(void) {
    const orc_resolution res[2] = {{0.5, 0.05, 20}, {1.0, 0.1, 30}};
    orc_map *m = orc_map_create(res, 2, 0.8);
    const size_t nmap = 60000;
    double *pts = (double *) malloc(nmap * 3 * sizeof(double));
    uint8_t *ins = (uint8_t *) malloc(nmap);
    for (size_t i = 0; i < nmap; ++i) room_point(pts + 3 * i);
    orc_map_insert(m, pts, nmap, ins);
    const double far_loc[3] = {4.0, 4.0, 0.0};
    orc_map_remove_far(m, far_loc, 9.0);
    orc_map_insert(m, pts, nmap / 2, ins);                       /* re-insert into tombstoned / reused storage */
    const uint64_t np = orc_map_num_points(m);
    double *exported = (double *) malloc((size_t) orc_map_export(m, 0, NULL, 0) * 3 * sizeof(double) + 24);
    const uint64_t ne = orc_map_export(m, 0, exported, orc_map_export(m, 0, NULL, 0));
    if (np == 0 || ne == 0 || orc_map_num_voxels(m, 0) == 0) { fprintf(stderr, "empty map\n"); return 2; }

    /* keypoints: room points seen from a slightly wrong pose */
    const size_t n = 3000;
    double *raw = (double *) malloc(n * 3 * sizeof(double)), *world = (double *) malloc(n * 3 * sizeof(double));
    double *t = (double *) malloc(n * sizeof(double));
    double pose[14] = {0, 0, 0.002, 1, 0.02, -0.01, 0.005, 0, 0, -0.002, 1, -0.015, 0.02, 0.0};
    const double tbe[2] = {0.0, 1.0};
    for (int e = 0; e < 2; ++e) orc_quat_normalize(pose + 7 * e);
    for (size_t i = 0; i < n; ++i) {
        room_point(raw + 3 * i);
        t[i] = (double) i / (double) (n - 1);
        orc_transform_point(pose, tbe, t[i], raw + 3 * i, world + 3 * i);
    }
    /* neighbour search, both tie rules, explicit and default radius, k at the cap */
    double nb[32 * 3];
    int total = 0;
    for (size_t i = 0; i < n; i += 7)
        for (int mode = 0; mode < 2; ++mode) {
            total += orc_map_radius_search(m, world + 3 * i, 0.0, 20, mode, nb);
            total += orc_map_radius_search(m, world + 3 * i, 1.3, 32, mode, nb);
        }
    if (total == 0) { fprintf(stderr, "no neighbours\n"); return 3; }
    /* GN: serial and threaded accumulation, full registration with the prior */
    orc_options o = {5, 20, 20, 0, 0.3, 0.0};
    orc_motion_prior pr = {0.001, 0.001, {0, 0, 0}, {0.01, 0, 0}};
    double A[144], b[12];
    int nu = 0;
    int32_t *nn = (int32_t *) malloc(n * sizeof(int32_t));
    double *normal = (double *) malloc(n * 3 * sizeof(double)), *a2d = (double *) malloc(n * sizeof(double)), *farp = (double *) malloc(n * 3 * sizeof(double));
    uint8_t *used = (uint8_t *) malloc(n);
    orc_gn_accumulate(m, raw, world, t, n, pose, tbe, &o, 0, 1, A, b, &nu, nn, normal, a2d, farp, used);
    orc_gn_accumulate(m, raw, world, t, n, pose, tbe, &o, 1, 4, A, b, &nu, NULL, NULL, NULL, NULL, NULL);
    orc_summary s;
    double pose_gn[14];
    memcpy(pose_gn, pose, sizeof(pose));
    double *w2 = (double *) malloc(n * 3 * sizeof(double));
    memcpy(w2, world, n * 3 * sizeof(double));
    if (orc_register_gn(m, raw, w2, t, n, pose_gn, tbe, &o, &pr, 0, 2, &s) != 0 || !s.success) { fprintf(stderr, "GN failed: %s\n", s.error_log); return 4; }
    /* soft failure (too few keypoints) and the timestamp error */
    if (orc_register_gn(m, raw, w2, t, 50, pose_gn, tbe, &o, NULL, 0, 1, &s) != 0 || s.success) { fprintf(stderr, "expected a soft failure\n"); return 5; }
    t[3] = 2.0;
    if (orc_register_gn(m, raw, w2, t, n, pose_gn, tbe, &o, NULL, 0, 1, &s) != -5) { fprintf(stderr, "expected the timestamp error\n"); return 6; }
    t[3] = 3.0 / (double) (n - 1);
    /* robust route, every loss */
    for (int loss = ORC_LOSS_STANDARD; loss <= ORC_LOSS_TRUNCATED; ++loss) {
        orc_robust_options ro = {3, 20, 20, 0, 900, loss, 3, 1, 0.9, 0.1, 2.0, 0.3, 0.1, 0.05, 0.0001, 0.001};
        orc_robust_prior rp = {0.001, 0.001, 0.0, 0.0, {0, 0, 0}, {0.01, 0, 0}, {0, 0, 0, 1}};
        double pose_r[14];
        memcpy(pose_r, pose, sizeof(pose));
        if (orc_register_robust(m, raw, w2, t, n, pose_r, tbe, &ro, &rp, 0, &s) != 0 || !s.success) { fprintf(stderr, "robust failed (loss %d): %s\n", loss, s.error_log); return 7; }
    }
    /* samplers + undistortion */
    uint32_t *idx = (uint32_t *) malloc(nmap * sizeof(uint32_t));
    const size_t k1 = orc_grid_sampling(pts, nmap, 0.5, idx);
    const double dist[6] = {0.5, 2.0, 4.0, 8.0, 16.0, 200.0}, vs[6] = {0.1, 0.2, 0.4, 0.8, 1.6, -1.0};
    const size_t k2 = orc_adaptive_sampling(pts, nmap, 1, 1500, 6, dist, vs, idx);
    double *und = (double *) malloc(n * 3 * sizeof(double));
    orc_transform_points(pose, tbe, t, raw, n, und, 3);
    if (k1 == 0 || k2 == 0 || k2 == (size_t) -1) { fprintf(stderr, "sampling failed\n"); return 8; }
    orc_map_clear(m);
    if (orc_map_num_points(m) != 0) return 9;
    orc_map_destroy(m);
    free(pts); free(ins); free(exported); free(raw); free(world); free(t); free(nn); free(normal); free(a2d); free(farp); free(used); free(w2); free(idx); free(und);
    printf("asan_oracle ok: %llu map points, %d neighbours visited, %zu / %zu sampled\n", (unsigned long long) np, total, k1, k2);
    return 0;
}
