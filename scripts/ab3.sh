#!/bin/bash
# round 3 A/B on ONE box: alternates ct_icp_amd/libctgn_base.so and ct_icp_amd/libctgn.so on the fresh-solve headline loop of bench.py
# (no extras, no sub-workloads, no PMC). Prints ms per step and the search-kernel time of first / later iterations per run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in $(seq 1 ${REPS:-2}); do
  for which in base cur; do
    lib="$PWD/ct_icp_amd/libctgn.so"; [ "$which" = "base" ] && lib="$PWD/ct_icp_amd/libctgn_base.so"
    CTGN_LIB_PATH=$lib python bench.py --steps ${STEPS:-100} --warmup 10 --clock-warm ${WARM:-100} --no-cpu-baseline --no-pmc --no-extras --sub none ${BENCH_ARGS:-} 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$which step_ms=%.4f kernel_ms=%.4f first=%.4f later=%.4f steady_step=%.4f steady_kernel=%.4f parity=%s' % (d['ms_per_step'], r['kernel_ms_avg'], r['first_iteration']['kernel_ms'], r['later_iterations']['kernel_ms'], r.get('steady_state_ms_per_step', 0), r.get('steady_state_kernel_ms_avg', 0), d.get('parity_m_rad')))"
  done
done
