#!/bin/bash
# round 4, GPU session 3: what bounds the pool check? k_pool_check (CTGN_SPLIT=1 forces the split launches on the B2 sweep) with its real
# scattered gathers and with the measurement hook that reads each pool as one contiguous run (bit 20: results invalid, time only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_3; mkdir -p $O
export TMPDIR=/tmp
for tag in real fake; do
  AB=0; [ $tag = fake ] && AB=1048576
  (cd /tmp && CTGN_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof_$tag" -o trace -- python "$OLDPWD/bench.py" --steps 100 --warmup 0 --inner --ablate $AB) > $O/rocprof_$tag.log 2>&1
  find $O/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -r cat > $O/kernel_stats_$tag.csv
  rm -rf $O/prof_$tag
  echo "== $tag"; cut -c1-200 $O/kernel_stats_$tag.csv | head -6
done
