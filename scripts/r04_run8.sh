#!/bin/bash
# round 4, GPU session 8: config E as defined on one GPU (profiles/r04_config_e_n1.json), the new GPU tests, the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_8; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 1200 python scripts/sequence_run.py --config-e --scale 10 --out $O/config_e_n1.json > /dev/null 2> $O/config_e.err
( time timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt; cp bench_detail.json $O/bench_default_detail.json
tail -n 6 $O/pytest_gpu.log; tail -n 12 $O/config_e.err; cut -c1-600 $O/config_e_n1.json; echo; cat $O/bench_default.json; cat $O/bench_default_time.txt
