#!/bin/bash
# round 4, GPU session 11: small frames — search + residual in one launch (k_search_residual; bit 23 = off). Parity suite; B1 / C step times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_11; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python scripts/iter_times.py B1 0 8388608 0 8388608 > $O/iter_B1.txt 2> $O/iter_B1.err
timeout 600 python scripts/iter_times.py C 0 8388608 0 8388608 > $O/iter_C.txt 2> $O/iter_C.err
grep -v "^  File" $O/pytest_gpu.log | tail -n 12 | cut -c1-300; cat $O/iter_B1.txt $O/iter_C.txt
