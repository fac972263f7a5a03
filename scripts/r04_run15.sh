#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_15; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for v in nofence fence; do
  CTGN_LIB_PATH=$PWD/ct_icp_amd/libctgn_$v.so timeout 600 python scripts/iter_times.py B2 0 > $O/iter_B2_${v}_$rep.txt 2> $O/err; echo "$v:"; cat $O/iter_B2_${v}_$rep.txt
done; done
for v in nofence fence; do CTGN_LIB_PATH=$PWD/ct_icp_amd/libctgn_$v.so timeout 600 python scripts/iter_times.py D 0 > $O/iter_D_$v.txt 2> $O/err; echo "D $v:"; cat $O/iter_D_$v.txt; done
