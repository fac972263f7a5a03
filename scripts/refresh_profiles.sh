#!/bin/bash
# The evidence of a round in one GPU-box session:  scripts/refresh_profiles.sh <round, e.g. 06>  -> gpurun_out/profiles_r<round>/ (copied to profiles/
# afterwards). Smoke + the -m gpu tier, the driver's bench command, rocprofv3 kernel traces (B2, D, B1), the five PMC passes (B2, D), the
# search-kernel time of each iteration, the search kernel's instruction budget by phase, the HIP API time line of a small-frame Register, and
# one config-E sequence through the reference's own Odometry (CPU map, GPU map, GPU map with the arms, + device shuffle) beside the library's loop.
# Replaces refresh_profiles_r02 .. r05.sh, refresh_bench_r03 / r04.sh and refresh_profiles_D.sh of the earlier rounds.
r=${1:?usage: refresh_profiles.sh <round>}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R="$PWD"; P=gpurun_out/profiles_r$r; rm -rf $P; mkdir -p $P
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(len(os.sched_getaffinity(0)))") > $P/r${r}_box.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -2 gpurun_out/smoke.log > $P/r${r}_smoke.txt; grep -E "passed|failed|rc=" gpurun_out/pytest_gpu.log >> $P/r${r}_smoke.txt
# the driver's command (defaults: headline B2 + sub-objects B1, C, D with D's counters, frame pipeline, config E / 10 + the reference's Odometry)
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $P/r${r}_bench_n1.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default_time.txt
cp bench_detail.json $P/r${r}_bench_detail.json; grep real gpurun_out/bench_default_time.txt > $P/r${r}_bench_n1_wallclock.txt
timeout 600 python bench.py --sub none --config-e-scale 0 --no-cpu-baseline > $P/r${r}_bench_n1_200steps.json 2> /dev/null
# the sharded loop on one rank through RCCL (a self all-reduce): the overheads a scaling record will carry
timeout 600 python bench.py --force-dist --workload D --steps 20 --warmup 2 --no-pmc --no-cpu-baseline --sub none --config-e-scale 0 > $P/r${r}_bench_D_sharded_loop_one_rank.json 2> /dev/null
# rocprofv3 kernel trace + stats of the timed loop
for w in B2 D B1; do
  rm -rf gpurun_out/prof$w
  st=200; [ $w = D ] && st=20
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof$w" -o trace -- python "$R/bench.py" --workload $w --steps $st --warmup 0 --inner) > gpurun_out/rocprof$w.log 2>&1
  out=$P/workload${w}_r${r}_rocprofv3_kernel_stats.csv; [ $w = B2 ] && out=$P/r${r}_rocprofv3_kernel_stats.csv
  find gpurun_out/prof$w -name "*kernel_stats.csv" | head -1 | xargs -r cat > $out
  rm -rf gpurun_out/prof$w
done
# PMC passes (separate runs, --pmc with --kernel-trace only)
for w in B2 D; do
  rm -f gpurun_out/pmc_?.txt
  BENCH_ARGS="--workload $w" scripts/gpu_pmc.sh > gpurun_out/pmc_all_$w.log 2>&1
  for i in 1 2 3 4 5; do
    out=$P/workload${w}_r${r}_pmc_pass$i.txt; [ $w = B2 ] && out=$P/r${r}_pmc_pass$i.txt
    [ -f gpurun_out/pmc_$i.txt ] && cp gpurun_out/pmc_$i.txt $out
  done
done
find gpurun_out/pmc? -type f -size +1M -delete 2>/dev/null
( for w in B2 D C B1; do timeout 400 python scripts/iter_times.py $w 0 2>&1 | grep '^{'; done ) > $P/r${r}_iter_times.txt
( for w in B2 D; do timeout 600 python scripts/rows_prof3.py $w 4 2>&1 | grep '^{'; done ) > $P/r${r}_search_kernel_phases.txt
# instruction budget of the search kernel by phase
for w in B2 D; do
  rm -rf gpurun_out/isa_tmp
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d "$R/gpurun_out/isa_tmp" -o pmc -- python "$R/scripts/isa_budget.py" run $w) > gpurun_out/isa_run_$w.log 2>&1
  grep '^{"workload"' gpurun_out/isa_run_$w.log > gpurun_out/isa_schedule_$w.json
  python scripts/isa_budget.py report gpurun_out/isa_schedule_$w.json $(find gpurun_out/isa_tmp -name "*counter_collection.csv" | head -1) > $P/r${r}_isa_budget_$w.txt 2>&1
  rm -rf gpurun_out/isa_tmp
done
# HIP API + kernel time line of a small-frame Register
for mode in gn robust; do
  rm -rf gpurun_out/api_tmp; flag=""; [ $mode = robust ] && flag="--robust"
  (cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$R/gpurun_out/api_tmp" -o t -- python "$R/scripts/register_api_trace.py" run $flag) > gpurun_out/api_run_$mode.log 2>&1
  python scripts/register_api_trace.py report $(find gpurun_out/api_tmp -name "*hip_api_trace.csv" | head -1) $(find gpurun_out/api_tmp -name "*kernel_trace.csv" | head -1) > $P/r${r}_register_hip_api_trace_$mode.txt 2>&1
  rm -rf gpurun_out/api_tmp
done
# one whole config-E sequence through the reference's own Odometry on its CPU map, on the GPU map, with the arms, + the library's own loop
timeout 2400 python tests/odometry_vs_reference.py --sequence 0 --frames 0 --solver GN --impl ref-cpu,ref-gpu,ref-gpu-armed,ref-gpu-armed-device-shuffle,ctgn --out $P/r${r}_config_e_seq0_vs_reference.json > /dev/null 2> gpurun_out/vs_reference.err
timeout 2400 python tests/odometry_vs_reference.py --sequence 0 --frames 200 --solver CERES --impl ref-gpu,ref-gpu-armed,ref-gpu-armed-device-shuffle --out $P/r${r}_config_e_seq0_vs_reference_ceres_200.json > /dev/null 2>> gpurun_out/vs_reference.err
ls -la $P; cat $P/r${r}_smoke.txt; cat $P/r${r}_bench_n1.json; cat $P/r${r}_bench_n1_wallclock.txt; cat $P/r${r}_iter_times.txt; tail -8 gpurun_out/vs_reference.err
