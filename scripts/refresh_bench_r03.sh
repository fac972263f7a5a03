cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R="$PWD"; P=gpurun_out/profiles_r03; mkdir -p $P
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1
grep '^{' gpurun_out/bench_default.log | tail -1 > $P/r03_bench_n1.json; grep real gpurun_out/bench_default.log > $P/r03_bench_n1_wallclock.txt
rm -rf gpurun_out/profD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/profD" -o trace -- python "$R/bench.py" --workload D --steps 20 --warmup 0 --inner) > gpurun_out/rocprofD.log 2>&1
find gpurun_out/profD -name "*kernel_stats.csv" | head -1 | xargs -r cat > $P/workloadD_r03_rocprofv3_kernel_stats.csv
( for w in B2 D C; do timeout 300 python scripts/iter_times.py $w 0 2048 2>&1 | grep '^{'; done ) > $P/r03_pools_iter_times.txt
timeout 300 python scripts/rows_prof3.py D 2 2>&1 | grep '^{' > $P/r03_search_kernel_phases_D.txt
find gpurun_out/profD -type f -size +1M -delete 2>/dev/null
cut -c1-300 $P/r03_bench_n1.json; cat $P/r03_pools_iter_times.txt
