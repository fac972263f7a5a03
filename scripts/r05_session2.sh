#!/bin/bash
# Round-5 session 2: pool margin A/B (B2, D), zero-copy end with path counters, kernel trace of B2 with / without the per-XCD pre-sums.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/s2; rm -rf $O; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=" $O/pytest_gpu.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sub none --config-e-scale 0"
line() { python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step %.4f kernel_ms %.4f first %.4f later %.4f certified %s parity %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["first_iteration"]["kernel_ms"], r["later_iterations"]["kernel_ms"], r["later_iterations"].get("pool_certified_frac"), d.get("parity_m_rad")))'; }
for m in 0 0.01 0.02 0.04 0 0.02; do
  echo "B2 pool_margin=$m: $(CTGN_TUNING=pool_margin=$m timeout 300 python bench.py $B 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_B2_margin.txt
for m in 0 0.02 0.04; do
  echo "D pool_margin=$m: $(CTGN_TUNING=pool_margin=$m timeout 400 python bench.py --workload D $B 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_D_margin.txt
timeout 300 python scripts/register_time.py B2 zero_copy_end=0,1 2>>$O/err.log | tee $O/ab_register.txt
for x in 1 0; do
  rm -rf gpurun_out/prof
  (cd /tmp && CTGN_TUNING=xcd_reduce=$x timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o trace -- python "$R/bench.py" --steps 100 --warmup 0 --inner) > $O/rocprof_$x.log 2>&1
  find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat | grep ctgn | sed -E 's/\(ctgn::[^"]*"/"/' | cut -c1-150 > $O/kernel_stats_xcd$x.csv
  cat $O/kernel_stats_xcd$x.csv | head -4
done
rm -rf gpurun_out/prof
for cfg in "B2 pool_margin=0" "B2 pool_margin=0.02" "D pool_margin=0" "D pool_margin=0.02"; do set -- $cfg; echo "$cfg: $(CTGN_TUNING=$2 timeout 400 python scripts/iter_times.py $1 0 2>&1 | grep '^{')"; done | tee $O/iter_times.txt
tail -5 $O/err.log
