#!/bin/bash
# Everything profiles/ holds, in one GPU-box session; results under gpurun_out/profiles_new/ (copy to profiles/ afterwards).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
P=gpurun_out/profiles_new; rm -rf $P; mkdir -p $P
VARIANTS="0" scripts/gpu_round.sh > gpurun_out/round.log 2>&1
grep '^{' gpurun_out/bench_v0.log | tail -1 > $P/r01_bench_n1.json
cp gpurun_out/kernel_stats.csv $P/r01_rocprofv3_kernel_stats.csv
cp gpurun_out/box.log $P/r01_box.txt
tail -2 gpurun_out/smoke.log > $P/r01_smoke.txt; grep -E "passed|failed" gpurun_out/pytest_gpu.log >> $P/r01_smoke.txt
scripts/gpu_pmc.sh > gpurun_out/pmc_all.log 2>&1
for i in 1 2 3 4 5; do cp gpurun_out/pmc_$i.txt $P/r01_pmc_pass$i.txt; done
python bench.py --variant 3 --no-pmc --no-cpu-baseline 2> $P/r01_phase_clocks.txt > /dev/null
BENCH_ARGS="--workload D" scripts/gpu_pmc.sh > gpurun_out/pmc_allD.log 2>&1
for i in 1 2 3 4 5; do cp gpurun_out/pmc_$i.txt $P/workloadD_r01_pmc_pass$i.txt; done
python bench.py --workload D --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $P/workloadD_r01_bench.json
python bench.py --workload D --no-cpu-baseline --no-pmc --order off 2>/dev/null | grep '^{' | tail -1 > $P/workloadD_r01_bench_unordered.json
python bench.py --no-cpu-baseline --no-pmc --order off 2>/dev/null | grep '^{' | tail -1 > $P/r01_bench_n1_unordered.json
python bench.py --workload B1 --no-cpu-baseline --no-pmc 2>/dev/null | grep '^{' | tail -1 > $P/r01_bench_B1.json
ls -la $P
