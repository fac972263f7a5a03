#!/bin/bash
# Round-5 session 4: gap certificate on / off (ablation bit 28), per-XCD pre-sums with the stamps read in one trip (kernel trace), B2 and D.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/s4; rm -rf $O; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|^E  |Error" $O/pytest_gpu.log | tail -12
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sub none --config-e-scale 0"
line() { python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step %.4f kernel_ms %.4f first %.4f later %.4f certified %s parity %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["first_iteration"]["kernel_ms"], r["later_iterations"]["kernel_ms"], r["later_iterations"].get("pool_certified_frac"), d.get("parity_m_rad")))'; }
G=268435456
for rep in 1 2; do
  echo "B2 gap-cert on , xcd on : $(timeout 300 python bench.py $B 2>>$O/err.log | line)"
  echo "B2 gap-cert off, xcd on : $(timeout 300 python bench.py $B --ablate $G 2>>$O/err.log | line)"
  echo "B2 gap-cert on , xcd off: $(CTGN_TUNING=xcd_reduce=0 timeout 300 python bench.py $B 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_B2.txt
for rep in 1 2; do
  echo "D gap-cert on : $(timeout 400 python bench.py --workload D $B 2>>$O/err.log | line)"
  echo "D gap-cert off: $(timeout 400 python bench.py --workload D $B --ablate $G 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_D.txt
rm -rf gpurun_out/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o trace -- python "$R/bench.py" --steps 100 --warmup 0 --inner) > $O/rocprof.log 2>&1
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat | grep ctgn | sed -E 's/\(ctgn::[^"]*"/"/' | cut -c1-150 > $O/kernel_stats.csv
head -4 $O/kernel_stats.csv
rm -rf gpurun_out/prof
for cfg in "B2 0" "B2 $G" "D 0" "D $G"; do set -- $cfg; echo "$cfg: $(timeout 400 python scripts/iter_times.py $1 $2 2>&1 | grep '^{')"; done | tee $O/iter_times.txt
tail -5 $O/err.log
