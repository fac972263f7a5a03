#!/bin/bash
# round 4, GPU session 1: parity suite after the exactness changes, A/B of the normals modes / first-search cull / residual block size,
# kernel trace of the B2 loop. Logs under gpurun_out/r04_1/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_1; mkdir -p $O
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null) > $O/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python scripts/iter_times.py B2 0 131072 196608 262144 0 > $O/iter_B2.txt 2> $O/iter_B2.err
CTGN_RES_BLOCK=256 timeout 600 python scripts/iter_times.py B2 0 196608 > $O/iter_B2_res256.txt 2> $O/iter_B2_res256.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o trace -- python "$OLDPWD/bench.py" --steps 200 --warmup 0 --inner) > $O/rocprof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r cat > $O/kernel_stats_B2.csv
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*.db" -delete
timeout 600 python scripts/iter_times.py D 0 > $O/iter_D.txt 2> $O/iter_D.err
tail -n 15 $O/pytest_gpu.log; tail -n 3 $O/smoke.log; cat $O/iter_B2.txt $O/iter_B2_res256.txt $O/iter_D.txt; cut -c1-150 $O/kernel_stats_B2.csv | head -12
