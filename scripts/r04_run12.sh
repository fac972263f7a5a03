#!/bin/bash
# round 4, GPU session 12: guessed first-search bound (bit 24 = off). Parity suite; D / B2 / C per-iteration times with and without it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_12; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 900 python scripts/iter_times.py D 0 16777216 0 16777216 > $O/iter_D.txt 2> $O/iter_D.err
CTGN_GUESS_FACTOR=1.3 timeout 900 python scripts/iter_times.py D 0 > $O/iter_D_f13.txt 2> $O/iter_D_f13.err
CTGN_GUESS_FACTOR=2.0 timeout 900 python scripts/iter_times.py D 0 > $O/iter_D_f20.txt 2> $O/iter_D_f20.err
timeout 600 python scripts/iter_times.py B2 0 16777216 > $O/iter_B2.txt 2> $O/iter_B2.err
grep -v "^  File" $O/pytest_gpu.log | tail -n 14 | cut -c1-300; cat $O/iter_D.txt; echo "factor 1.3:"; cat $O/iter_D_f13.txt; echo "factor 2.0:"; cat $O/iter_D_f20.txt; cat $O/iter_B2.txt
