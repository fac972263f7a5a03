#!/bin/bash
# Kernel trace + PMC passes of the headline workload on the round's FINAL build (the evidence session predates the last adopted change).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R="$PWD"; P=gpurun_out/final3; rm -rf $P; mkdir -p $P
rm -rf gpurun_out/profB2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/profB2" -o trace -- python "$R/bench.py" --workload B2 --steps 200 --warmup 0 --inner) > gpurun_out/rocprofB2.log 2>&1
find gpurun_out/profB2 -name "*kernel_stats.csv" | head -1 | xargs -r cat > $P/r05_rocprofv3_kernel_stats.csv; rm -rf gpurun_out/profB2
rm -f gpurun_out/pmc_?.txt
BENCH_ARGS="--workload B2" scripts/gpu_pmc.sh > gpurun_out/pmc_all_B2.log 2>&1
for i in 1 2 3 4 5; do [ -f gpurun_out/pmc_$i.txt ] && cp gpurun_out/pmc_$i.txt $P/r05_pmc_pass$i.txt; done
find gpurun_out/pmc? -type f -size +1M -delete 2>/dev/null
timeout 400 python scripts/iter_times.py B2 0 2>&1 | grep '^{' > $P/r05_iter_times_B2.txt
ls $P; cut -c1-150 $P/r05_rocprofv3_kernel_stats.csv | head -5; cat $P/r05_iter_times_B2.txt; head -12 $P/r05_pmc_pass1.txt
