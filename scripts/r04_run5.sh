#!/bin/bash
# round 4, GPU session 5: neighbour blocks (CACHED_FLAG). Parity suite; A/B through bit 21 of the ablation mask (blocks off) on B2 and D;
# kernel trace of the B2 loop.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python scripts/iter_times.py B2 0 2097152 0 2097152 > $O/iter_B2.txt 2> $O/iter_B2.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o trace -- python "$OLDPWD/bench.py" --steps 200 --warmup 0 --inner) > $O/rocprof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat > $O/kernel_stats_B2.csv
rm -rf $O/prof
timeout 600 python scripts/iter_times.py D 0 2097152 524288 > $O/iter_D.txt 2> $O/iter_D.err
tail -n 12 $O/pytest_gpu.log; cat $O/iter_B2.txt $O/iter_D.txt; cut -c1-170 $O/kernel_stats_B2.csv | head -8
