#!/bin/bash
# Round-5 session 10: pools kept by searches whose carried-over bound lies beyond the radius (ablation bit 29 = as before): parity tier, B2, D, C.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/s10; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|^E  |Error" $O/pytest_gpu.log | tail -8
G=536870912
for cfg in "B2 0" "B2 $G" "B2 0" "B2 $G" "D 0" "D $G"; do set -- $cfg; echo "$cfg: $(timeout 400 python scripts/iter_times.py $1 $2 2>&1 | grep '^{')"; done | tee $O/iter_times.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sub none --config-e-scale 0"
line() { python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step %.4f kernel_ms %.4f first %.4f later %.4f certified %s parity %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["first_iteration"]["kernel_ms"], r["later_iterations"]["kernel_ms"], r["later_iterations"].get("pool_certified_frac"), d.get("parity_m_rad")))'; }
for w in B2 C B1; do echo "$w on : $(timeout 300 python bench.py --workload $w $B 2>>$O/err.log | line)"; echo "$w off: $(timeout 300 python bench.py --workload $w $B --ablate $G 2>>$O/err.log | line)"; done 2>&1 | tee $O/ab.txt
tail -3 $O/err.log
