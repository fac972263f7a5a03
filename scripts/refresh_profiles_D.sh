#!/bin/bash
# workload-D part of profiles/ (see refresh_profiles.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
P=gpurun_out/profiles_new; mkdir -p $P
BENCH_ARGS="--workload D" scripts/gpu_pmc.sh > gpurun_out/pmc_allD.log 2>&1
for i in 1 2 3 4 5; do cp gpurun_out/pmc_$i.txt $P/workloadD_r01_pmc_pass$i.txt; done
python bench.py --workload D --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $P/workloadD_r01_bench.json
python bench.py --workload D --no-cpu-baseline --no-pmc --order off 2>/dev/null | grep '^{' | tail -1 > $P/workloadD_r01_bench_unordered.json
cut -c1-260 $P/workloadD_r01_bench.json; echo; cat $P/workloadD_r01_pmc_pass4.txt $P/workloadD_r01_pmc_pass5.txt | head -8
