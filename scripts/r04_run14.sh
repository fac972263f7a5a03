#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_14; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/iter_times.py B2 0 > $O/iter_B2_off.txt 2> $O/err1; echo "B2 guess off (default rule):"; cat $O/iter_B2_off.txt
for f in 1.1 1.25 1.4; do CTGN_GUESS_MAXFRAC=0.95 CTGN_GUESS_FACTOR=$f timeout 600 python scripts/iter_times.py B2 0 > $O/iter_B2_f$f.txt 2> $O/err2; echo "B2 forced factor $f:"; cat $O/iter_B2_f$f.txt; done
CTGN_GUESS_MAXFRAC=0.95 CTGN_GUESS_FACTOR=1.25 timeout 600 python scripts/iter_times.py C 0 16777216 > $O/iter_C.txt 2> $O/err3; echo "C forced 1.25 / off:"; cat $O/iter_C.txt
