// asan_host_mirror.cpp — the PRODUCT's host-side map mirror (ct_icp_amd/csrc/ctgn_map.hpp: insert rule, eviction, tombstones, block reuse,
// rehash / growth, export, edit log) fuzzed against a plain std::map model under -fsanitize=address,undefined. Built host-only by hipcc
// (`make -C oracle asan`), run by tests/test_sanitizers.py. No device code, no oracle: this checks memory safety and the map rules of
// include/ct_icp/map.h:261-293,305-322 on the host mirror itself.
#include <array>
#include <cstdio>
#include <map>
#include <random>
#include <vector>

#include "ctgn_map.hpp"

using namespace ctgn;

int main() {
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    std::normal_distribution<double> nrm(0.0, 1.0);
    const double res = 0.5, mind = 0.07;
    const int blk = 12;
    VoxelLevel L;
    L.init(res, mind, blk, 0);
    L.log_edits = true;
    L.need_full_upload = false;                       // exercise the edit log as after a first device sync
    std::map<std::array<int, 3>, std::vector<std::array<double, 3>>> model;
    size_t inserted = 0, model_points = 0;
    for (int round = 0; round < 40; ++round) {
        const double cx = 8.0 * std::cos(0.3 * round), cy = 8.0 * std::sin(0.3 * round);
        for (int i = 0; i < 6000; ++i) {
            const double p[3] = {cx + 3.0 * nrm(g), cy + 3.0 * nrm(g), 0.4 * nrm(g)};
            const int r = L.insert_point(p[0], p[1], p[2]);
            std::array<int, 3> key{(int) (p[0] / res), (int) (p[1] / res), (int) (p[2] / res)};
            auto &vox = model[key];
            bool take = vox.empty();
            if (!take && (int) vox.size() < blk) {
                double sq_min = 1e300;
                for (auto &q : vox) sq_min = std::min(sq_min, sq_norm3(q[0] - p[0], q[1] - p[1], q[2] - p[2]));
                take = sq_min > mind * mind;
            }
            if (take) { vox.push_back({p[0], p[1], p[2]}); ++model_points; }
            if ((r == 1) != take) { std::fprintf(stderr, "insert decision differs at round %d point %d\n", round, i); return 2; }
            inserted += r == 1;
        }
        if (round % 3 == 2) {                          // evict far voxels on the first point's distance
            const double loc[3] = {cx, cy, 0.0};
            L.remove_far(loc, 7.0);
            for (auto it = model.begin(); it != model.end();) {
                const auto &f = it->second.front();
                const double d = std::sqrt(sq_norm3(f[0] - loc[0], f[1] - loc[1], f[2] - loc[2]));
                if (d > 7.0) { model_points -= it->second.size(); it = model.erase(it); } else ++it;
            }
        }
        if (L.num_points != model_points || L.num_voxels != model.size()) {
            std::fprintf(stderr, "counts differ after round %d: %llu / %zu points, %llu / %zu voxels\n", round, (unsigned long long) L.num_points, model_points,
                         (unsigned long long) L.num_voxels, model.size());
            return 3;
        }
        if (round % 5 == 4) { L.slot_edits.clear(); L.point_edits.clear(); }     // a device sync would consume the log
    }
    std::vector<double> out(3 * L.num_points + 3);
    const uint64_t n = L.export_points(out.data(), L.num_points);
    if (n != L.num_points) return 4;
    L.clear();
    if (L.num_points != 0) return 5;
    std::printf("asan_host_mirror ok: %zu inserted, %zu points in %zu voxels at the end\n", inserted, model_points, model.size());
    return 0;
}
