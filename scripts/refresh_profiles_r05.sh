#!/bin/bash
# Round-5 evidence in one GPU-box session; results under gpurun_out/profiles_r05/ (copied to profiles/ afterwards).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R="$PWD"; P=gpurun_out/profiles_r05; rm -rf $P; mkdir -p $P
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(len(os.sched_getaffinity(0)))") > $P/r05_box.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -2 gpurun_out/smoke.log > $P/r05_smoke.txt; grep -E "passed|failed|rc=" gpurun_out/pytest_gpu.log >> $P/r05_smoke.txt
# the driver's command (defaults: headline B2 + sub-objects B1, C, D with D's counters, frame pipeline, config E / 10)
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $P/r05_bench_n1.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default_time.txt
cp bench_detail.json $P/r05_bench_detail.json; grep real gpurun_out/bench_default_time.txt > $P/r05_bench_n1_wallclock.txt
# the same line with the round's usual K (200 steps)
timeout 600 python bench.py --sub none --config-e-scale 0 --no-cpu-baseline > $P/r05_bench_n1_200steps.json 2> /dev/null
# rocprofv3 kernel trace + stats of the timed loop: default workload, D, B1
for w in B2 D B1; do
  rm -rf gpurun_out/prof$w
  st=200; [ $w = D ] && st=20
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof$w" -o trace -- python "$R/bench.py" --workload $w --steps $st --warmup 0 --inner) > gpurun_out/rocprof$w.log 2>&1
  out=$P/workload${w}_r05_rocprofv3_kernel_stats.csv; [ $w = B2 ] && out=$P/r05_rocprofv3_kernel_stats.csv
  find gpurun_out/prof$w -name "*kernel_stats.csv" | head -1 | xargs -r cat > $out
  rm -rf gpurun_out/prof$w
done
# PMC passes (separate runs, --pmc with --kernel-trace only): B2, D
for w in B2 D; do
  rm -f gpurun_out/pmc_?.txt
  BENCH_ARGS="--workload $w" scripts/gpu_pmc.sh > gpurun_out/pmc_all_$w.log 2>&1
  for i in 1 2 3 4 5; do
    out=$P/workload${w}_r05_pmc_pass$i.txt; [ $w = B2 ] && out=$P/r05_pmc_pass$i.txt
    [ -f gpurun_out/pmc_$i.txt ] && cp gpurun_out/pmc_$i.txt $out
  done
done
# search-kernel time of each iteration of fresh solves
( for w in B2 D C B1; do timeout 400 python scripts/iter_times.py $w 0 2>&1 | grep '^{'; done ) > $P/r05_iter_times.txt
find gpurun_out/pmc? -type f -size +1M -delete 2>/dev/null
ls -la $P; cat $P/r05_smoke.txt; cat $P/r05_bench_n1.json; cat $P/r05_bench_n1_wallclock.txt; cat $P/r05_iter_times.txt; tail -3 gpurun_out/bench_default.err
