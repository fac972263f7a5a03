#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_13; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "guess or config_e or shared or pools" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
for f in 0.9 1.05 1.15; do CTGN_GUESS_FACTOR=$f timeout 900 python scripts/iter_times.py D 0 > $O/iter_D_f$f.txt 2> $O/iter_D_f$f.err; echo "factor $f:"; cat $O/iter_D_f$f.txt; done
grep -v "^  File" $O/pytest_gpu.log | tail -n 8 | cut -c1-300
