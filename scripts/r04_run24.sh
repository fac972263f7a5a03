#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_24; mkdir -p $O
export TMPDIR=/tmp
for cap in 0 1024 768 1536; do CTGN_RES_GRID_CAP=$cap timeout 600 python scripts/iter_times.py D 0 > $O/iter_D_cap$cap.txt 2> $O/err; echo "D residual grid cap $cap:"; cat $O/iter_D_cap$cap.txt; done
