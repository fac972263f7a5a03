// Sanitizer harness for the product's host-side helper-thread pool (ct_icp_amd/csrc/ctgn_hostpool.hpp): test infrastructure only.
// Built twice by `make -C oracle asan` — under -fsanitize=thread and under -fsanitize=address,undefined — and run by tests/test_sanitizers.py.
// Checks the contract of HostPool::run: every part exactly once, all done on return, no job touched after its run() returned (the
// closures live on this stack frame), with 0..5 helpers, part counts below / at / far above the thread count, back-to-back runs.
#include <cstdio>
#include <numeric>

#include "ctgn_hostpool.hpp"

int main() {
    unsigned long long checksum = 0;
    for (int helpers = 0; helpers <= 5; ++helpers) {
        ctgn::HostPool pool;
        pool.ensure(helpers);
        pool.ensure(helpers);                       // idempotent
        for (int rep = 0; rep < 300; ++rep) {
            const size_t parts = (size_t) (rep % 7 == 0 ? 0 : rep % 5 == 0 ? 1 : 1 + (rep * 37) % 61);
            std::vector<int> hits(parts, 0);
            std::vector<unsigned long long> out(parts, 0);
            const std::function<void(size_t)> job = [&](size_t i) {
                ++hits[i];                          // a part run twice, or by two threads at once, shows here (and to the sanitizer)
                unsigned long long a = 0;
                for (size_t k = 0; k < 200 + 13 * i; ++k) a += k * (i + 1);
                out[i] = a;
            };
            pool.run(parts, job);
            for (size_t i = 0; i < parts; ++i) {
                if (hits[i] != 1) { std::printf("part %zu ran %d times (helpers %d, parts %zu)\n", i, hits[i], helpers, parts); return 1; }
                checksum += out[i];
            }
        }
    }
    std::printf("hostpool ok %llu\n", checksum);
    return 0;
}
