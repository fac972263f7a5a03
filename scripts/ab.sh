#!/bin/bash
# A/B timing on ONE box (box-to-box noise is +-4%): alternates the baseline build (ct_icp_amd/libctgn_base.so) and the
# current build (ct_icp_amd/libctgn.so), REPS times each.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in $(seq 1 ${REPS:-3}); do
  for which in base cur; do
    lib="$PWD/ct_icp_amd/libctgn.so"; [ "$which" = "base" ] && lib="$PWD/ct_icp_amd/libctgn_base.so"
    CTGN_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc ${BENCH_ARGS:-} 2>/dev/null | tail -1 | sed -E "s/.*\"ms_per_step\": ([0-9.]+).*\"kernel_ms_avg\": ([0-9.]+).*/$which step_ms=\1 kernel_ms=\2/"
  done
done
