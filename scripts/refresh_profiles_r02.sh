#!/bin/bash
# Round-2 evidence in one GPU-box session; results under gpurun_out/profiles_r02/ (copied to profiles/ afterwards).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R="$PWD"; P=gpurun_out/profiles_r02; rm -rf $P; mkdir -p $P
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os;print(len(os.sched_getaffinity(0)))") > $P/r02_box.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -2 gpurun_out/smoke.log > $P/r02_smoke.txt; grep -E "passed|failed|rc=" gpurun_out/pytest_gpu.log >> $P/r02_smoke.txt
# the driver's command (defaults), then the other workloads
( time timeout 600 python bench.py ) > gpurun_out/bench_default.log 2>&1
grep '^{' gpurun_out/bench_default.log | tail -1 > $P/r02_bench_n1.json; grep real gpurun_out/bench_default.log > $P/r02_bench_n1_wallclock.txt
for w in B2-small B1 D; do timeout 400 python bench.py --workload $w --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > $P/r02_bench_$w.json; done
# rocprofv3 kernel trace + stats of the timed loop of the default workload
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o trace -- python "$R/bench.py" --steps 200 --warmup 0 --inner) > gpurun_out/rocprof.log 2>&1
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat > $P/r02_rocprofv3_kernel_stats.csv
# PMC passes (separate runs, --pmc with --kernel-trace only)
scripts/gpu_pmc.sh > gpurun_out/pmc_all.log 2>&1
for i in 1 2 3 4 5; do cp gpurun_out/pmc_$i.txt $P/r02_pmc_pass$i.txt; done
# phase ablations of the search kernel (results invalid, timings only)
( echo "# B2"; MASKS="0 2 256 512 32 16 1024" bash scripts/ablate2.sh; echo "# B2-small"; EXTRA="--workload B2-small" MASKS="0 2 256 512 32 16 1024" bash scripts/ablate2.sh ) > $P/r02_search_kernel_ablation.txt 2>&1
# frame pipeline: host-clock marks + kernel trace
timeout 300 python scripts/frame_prof.py 2>&1 | grep -v amdgpu.ids | tail -14 > $P/r02_frame_pipeline_marks.txt
rm -rf gpurun_out/fprof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/fprof" -o fp -- python "$R/scripts/frame_prof.py") > /dev/null 2>&1
find gpurun_out/fprof -name "*kernel_stats.csv" | head -1 | xargs -r cat > $P/r02_frame_pipeline_kernel_stats.csv
# whole sequences
timeout 300 python scripts/sequence_run.py --frames 40 --pipeline 2>/dev/null | tail -1 > $P/r02_sequence_gn_pipeline.json
timeout 300 python scripts/sequence_run.py --frames 40 2>/dev/null | tail -1 > $P/r02_sequence_gn_stage_calls.json
timeout 300 python scripts/sequence_run.py --frames 40 --pipeline --solver CERES 2>/dev/null | tail -1 > $P/r02_sequence_ceres_pipeline.json
timeout 300 python scripts/sequence_run.py --frames 20 --sequences 2 --pipeline 2>/dev/null | tail -1 > $P/r02_sequence_two_sequences_one_gpu.json
find gpurun_out/prof gpurun_out/fprof gpurun_out/pmc? -type f -size +1M -delete 2>/dev/null
ls -la $P; cat $P/r02_smoke.txt; cut -c1-600 $P/r02_bench_n1.json
