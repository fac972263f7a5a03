#!/bin/bash
# Ablation timing of the row kernel with warm clocks: kernel ms over 40 steps with phases skipped (results invalid, timings only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in ${MASKS:-0 32 1 33 2 16}; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --ablate $m ${EXTRA} 2>/dev/null | tail -1 | sed -E "s/.*\"kernel_ms_avg\": ([0-9.]+).*/ablate=$m kernel_ms=\1/"
done
