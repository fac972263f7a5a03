#!/bin/bash
# Round-3 evidence in one GPU-box session; results under gpurun_out/profiles_r03/ (copied to profiles/ afterwards).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R="$PWD"; P=gpurun_out/profiles_r03; rm -rf $P; mkdir -p $P
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(len(os.sched_getaffinity(0)))") > $P/r03_box.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -2 gpurun_out/smoke.log > $P/r03_smoke.txt; grep -E "passed|failed|rc=" gpurun_out/pytest_gpu.log >> $P/r03_smoke.txt
# the driver's command (defaults: headline B2 + sub-objects B1, C, D)
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1
grep '^{' gpurun_out/bench_default.log | tail -1 > $P/r03_bench_n1.json; grep real gpurun_out/bench_default.log > $P/r03_bench_n1_wallclock.txt
timeout 400 python bench.py --workload B2-small --no-cpu-baseline --no-extras --sub none 2>/dev/null | grep '^{' | tail -1 > $P/r03_bench_B2-small.json
# rocprofv3 kernel trace + stats of the timed loop of the default workload (fresh 5-iteration solves)
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o trace -- python "$R/bench.py" --steps 200 --warmup 0 --inner) > gpurun_out/rocprof.log 2>&1
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat > $P/r03_rocprofv3_kernel_stats.csv
# the same for workload D (one GPU) and for the small-frame regime (B1: the persistent kernel)
rm -rf gpurun_out/profD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/profD" -o trace -- python "$R/bench.py" --workload D --steps 20 --warmup 0 --inner) > gpurun_out/rocprofD.log 2>&1
find gpurun_out/profD -name "*kernel_stats.csv" | head -1 | xargs -r cat > $P/workloadD_r03_rocprofv3_kernel_stats.csv
rm -rf gpurun_out/profB1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/profB1" -o trace -- python "$R/bench.py" --workload B1 --steps 200 --warmup 0 --inner) > gpurun_out/rocprofB1.log 2>&1
find gpurun_out/profB1 -name "*kernel_stats.csv" | head -1 | xargs -r cat > $P/workloadB1_r03_rocprofv3_kernel_stats.csv
# PMC passes (separate runs, --pmc with --kernel-trace only)
scripts/gpu_pmc.sh > gpurun_out/pmc_all.log 2>&1
for i in 1 2 3 4 5; do cp gpurun_out/pmc_$i.txt $P/r03_pmc_pass$i.txt; done
# phase clocks of the search kernel on fresh solves: first search of a solve vs the bounded ones
timeout 300 python scripts/rows_prof3.py B2 6 2>&1 | grep '^{' > $P/r03_search_kernel_phases.txt
# neighbour pools: search-kernel time of each iteration of fresh solves, pools on (mask 0) and off (mask 2048)
( for w in B2 D C; do timeout 300 python scripts/iter_times.py $w 0 2048 2>&1 | grep '^{'; done ) > $P/r03_pools_iter_times.txt
# small frames: where a fresh solve spends its time, persistent kernel on and off
( for p in 1 0; do echo "CTGN_PERSISTENT=$p"; CTGN_PERSISTENT=$p timeout 300 python scripts/fresh_probe.py B1 2>&1 | grep -E "back-to-back|device stamps|us host" | tail -3; done ) > $P/r03_small_frame_probe.txt 2>&1
# frame pipeline: host-clock marks
timeout 300 python scripts/frame_prof.py 2>&1 | grep -v amdgpu.ids | tail -14 > $P/r03_frame_pipeline_marks.txt
# whole sequences
timeout 300 python scripts/sequence_run.py --frames 40 --pipeline 2>/dev/null | tail -1 > $P/r03_sequence_gn_pipeline.json
timeout 300 python scripts/sequence_run.py --frames 40 --pipeline --solver CERES 2>/dev/null | tail -1 > $P/r03_sequence_ceres_pipeline.json
# two ranks on the one GPU (gloo): the N > 1 path of bench.py
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --dist-backend gloo --steps 10 --warmup 2 --clock-warm 5 --d-sweeps 2 --d-radius 60 --no-pmc 2>/dev/null | grep '^{' | tail -1 > $P/r03_bench_n2_gloo_rehearsal.json
find gpurun_out/prof gpurun_out/profD gpurun_out/profB1 gpurun_out/pmc? -type f -size +1M -delete 2>/dev/null
ls -la $P; cat $P/r03_smoke.txt; cut -c1-400 $P/r03_bench_n1.json
