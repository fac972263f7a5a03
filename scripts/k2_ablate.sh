#!/bin/bash
# Ablation of k_residual_reduce on one box: rocprofv3 average of the kernel with phases skipped (results invalid while a mask is set).
# masks: 0 none, 4 no gathers/sums, 8 no eigen-solve/residual/Jacobian, 128 no packed accumulation, 140 all three
cd "$(dirname "$0")/.."; R="$PWD"; export TMPDIR=/tmp
for m in 0 4 8 128 140; do
  rm -rf /tmp/k2ab; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2ab -o t -- python "$R/bench.py" --inner --steps 10 --warmup 2 --ablate $m) > /dev/null 2>&1
  f=$(find /tmp/k2ab -name "*kernel_stats.csv" | head -1)
  echo "ablate=$m $(grep k_residual_reduce "$f" | awk -F, '{print "k_residual_reduce avg_ns=" $(NF-4)}') $(grep k_reduce_solve "$f" | awk -F, '{print "k_reduce_solve avg_ns=" $(NF-4)}')"
done
