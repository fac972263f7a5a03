#!/bin/bash
# round 3, first box session: smoke, the new GPU tests, the default bench line (new fresh-solve headline + sub-workloads)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; free -g | head -2) > gpurun_out/box.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "rewind or library_side or rehearsal or stepwise or sharded_loop" > gpurun_out/pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_new.log
( time timeout 900 python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?" >> gpurun_out/bench_default.err
tail -n 3 gpurun_out/smoke.log; tail -n 25 gpurun_out/pytest_new.log; tail -n 12 gpurun_out/bench_default.err; cut -c1-3000 gpurun_out/bench_default.json
