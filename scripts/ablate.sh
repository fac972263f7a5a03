#!/bin/bash
# Ablation timing of the row kernel on one box: first-launch kernel ms with phases skipped (results invalid, timings only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in ${MASKS:-0 2 4 8 6 14 15 1 17 0}; do
  for rep in 1 2 3; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --ablate $m 2>/dev/null | tail -1 | sed -E "s/.*\"kernel_ms_avg\": ([0-9.]+).*/ablate=$m kernel_ms=\1/"
  done
done
