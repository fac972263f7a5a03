#!/bin/bash
# round 4, GPU session 17: full parity suite + smoke + the default bench line as the driver runs it (with the D counters).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_17; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt; cp bench_detail.json $O/bench_default_detail.json
grep -v "^  File" $O/pytest_gpu.log | tail -n 8 | cut -c1-300; tail -2 $O/smoke.log; cat $O/bench_default.json; tail -4 $O/bench_default_time.txt; tail -3 $O/bench_default.err
