#!/bin/bash
# Round-5 session 5: the robust route's fused evaluation + step launch (k_robust_eval_step) on / off.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/s5; rm -rf $O; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|^E  |Error" $O/pytest_gpu.log | tail -12
for rep in 1 2; do for f in 0 1; do
  echo "robust_fuse=$f: $(CTGN_TUNING=robust_fuse=$f timeout 300 python scripts/robust_bench.py --reps 40 2>>$O/err.log | grep -v amdgpu | tail -3 | tr '\n' ' ')"
done; done 2>&1 | tee $O/ab_robust.txt
rm -rf gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o trace -- python "$R/scripts/robust_bench.py" --reps 40) > $O/rocprof.log 2>&1
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat | grep ctgn | sed -E 's/\(ctgn::[^"]*"/"/' | cut -c1-150 > $O/kernel_stats_robust.csv
head -12 $O/kernel_stats_robust.csv
rm -rf gpurun_out/prof
tail -5 $O/err.log
