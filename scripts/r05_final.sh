#!/bin/bash
# Round-5 closing session on the final build: smoke, the whole -m gpu tier, the driver's bench command (the committed line), config E on
# both routes, the phase clocks of config D's searches. Results under gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -2 $O/smoke.log | grep -v amdgpu > $O/r05_smoke.txt; grep -E "passed|failed|rc=" $O/pytest_gpu.log >> $O/r05_smoke.txt; cat $O/r05_smoke.txt
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r05_bench_n1.json 2> $O/bench_default.err ) 2> $O/bench_default_time.txt
cp bench_detail.json $O/r05_bench_detail.json; grep real $O/bench_default_time.txt > $O/r05_bench_n1_wallclock.txt
timeout 600 python bench.py --sub none --config-e-scale 0 --no-cpu-baseline > $O/r05_bench_n1_200steps.json 2> /dev/null
timeout 1500 python scripts/sequence_run.py --config-e --scale 10 --solver GN,CERES --out $O/r05_config_e_n1.json > /dev/null 2> $O/r05_config_e_log.txt
for w in D B2; do timeout 400 python scripts/rows_prof3.py $w 3 2>&1 | grep '^{'; done > $O/r05_search_kernel_phases.txt
cut -c1-700 $O/r05_bench_n1.json; cat $O/r05_bench_n1_wallclock.txt; grep -v amdgpu $O/r05_config_e_log.txt | tail -4; cut -c1-900 $O/r05_search_kernel_phases.txt; tail -3 $O/bench_default.err
