#!/bin/bash
# Round-5 session 8: pool checks that cannot pass unless the k-th distance shrank are skipped (KpView::check_skip) — B2 A/B, parity under it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/s8; rm -rf $O; mkdir -p $O
CTGN_TUNING=check_skip=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -p no:cacheprovider -k "pools or full_size or km_scale or guessed or config_d or wave_shared" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc=|^E  |Error" $O/pytest_gpu.log | tail -8
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sub none --config-e-scale 0"
line() { python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step %.4f kernel_ms %.4f first %.4f later %.4f certified %s parity %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["first_iteration"]["kernel_ms"], r["later_iterations"]["kernel_ms"], r["later_iterations"].get("pool_certified_frac"), d.get("parity_m_rad")))'; }
for c in 0 1 0.5 2 0 1 4; do
  echo "B2 check_skip=$c: $(CTGN_TUNING=check_skip=$c timeout 300 python bench.py $B 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_B2.txt
for c in 0 1; do echo "B2 check_skip=$c: $(CTGN_TUNING=check_skip=$c timeout 400 python scripts/iter_times.py B2 0 2>&1 | grep '^{')"; done | tee $O/iter_times.txt
tail -3 $O/err.log
