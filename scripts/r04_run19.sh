#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_19; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "sort or config_d or ordering or sharding or frame or device_resident or sampling" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 900 python scripts/iter_times.py D 0 > $O/iter_D.txt 2> $O/err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/profD" -o trace -- python "$OLDPWD/bench.py" --workload D --steps 20 --warmup 0 --inner) > $O/rocprofD.log 2>&1
find $O/profD -name "*kernel_stats.csv" | head -1 | xargs -r cat > $O/kernel_stats_D.csv; rm -rf $O/profD
grep -v "^  File" $O/pytest_gpu.log | tail -n 5 | cut -c1-300; cat $O/iter_D.txt; grep "ctgn::" $O/kernel_stats_D.csv | sed 's/(ctgn::MapView.*"/"/; s/(unsigned.*"/"/; s/(double.*"/"/' | cut -c1-150
