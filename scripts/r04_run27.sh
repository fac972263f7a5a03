#!/bin/bash
# session 27: pool size (members kept behind the k neighbours: ablation-mask bits 12-15 override POOL_EXTRA = 8) on D and B2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_27; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/iter_times.py D 0 49152 40960 0 49152 > $O/iter_D.txt 2> $O/err; cat $O/iter_D.txt
timeout 600 python scripts/iter_times.py B2 0 49152 40960 24576 0 > $O/iter_B2.txt 2>> $O/err; cat $O/iter_B2.txt
tail -2 $O/err
