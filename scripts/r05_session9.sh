#!/bin/bash
# Round-5 session 9: check_skip from the 4th / 5th search of a solve on (B2, D).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/s9; rm -rf $O; mkdir -p $O
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sub none --config-e-scale 0"
line() { python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step %.4f kernel_ms %.4f first %.4f later %.4f certified %s parity %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["first_iteration"]["kernel_ms"], r["later_iterations"]["kernel_ms"], r["later_iterations"].get("pool_certified_frac"), d.get("parity_m_rad")))'; }
for cfg in check_skip=0 check_skip=1 check_skip=2 check_skip=0 check_skip=1 check_skip=1,check_skip_from=5 check_skip=4,check_skip_from=5 check_skip=0.5,check_skip_from=3; do
  echo "B2 $cfg: $(CTGN_TUNING=$cfg timeout 300 python bench.py $B 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_B2.txt
for cfg in check_skip=0 check_skip=1 check_skip=2; do echo "B2 $cfg: $(CTGN_TUNING=$cfg timeout 400 python scripts/iter_times.py B2 0 2>&1 | grep '^{')"; done | tee $O/iter_times.txt
tail -3 $O/err.log
