#!/bin/bash
# Round-5 session 7: the gap certificate inside k_pool_check only (the split launches of frames >= 400 k keypoints): D on / off.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/s7; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|^E  |Error" $O/pytest_gpu.log | tail -12
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sub none --config-e-scale 0"
line() { python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step %.4f kernel_ms %.4f first %.4f later %.4f parity %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["first_iteration"]["kernel_ms"], r["later_iterations"]["kernel_ms"], d.get("parity_m_rad")))'; }
G=268435456
for rep in 1 2; do
  echo "D gap-cert on : $(timeout 400 python bench.py --workload D $B 2>>$O/err.log | line)"
  echo "D gap-cert off: $(timeout 400 python bench.py --workload D $B --ablate $G 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_D.txt
echo "B2 (untouched kernels): $(timeout 300 python bench.py $B 2>>$O/err.log | line)" | tee $O/ab_B2.txt
for cfg in "D 0" "D $G"; do set -- $cfg; echo "$cfg: $(timeout 400 python scripts/iter_times.py $1 $2 2>&1 | grep '^{')"; done | tee $O/iter_times.txt
tail -3 $O/err.log
