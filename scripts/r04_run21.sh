#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_21; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for v in ctgn ctgn_nu; do
  CTGN_LIB_PATH=$PWD/ct_icp_amd/lib$v.so timeout 600 python scripts/iter_times.py D 0 > $O/iter_D_${v}_$rep.txt 2> $O/err; echo "D $v:"; cat $O/iter_D_${v}_$rep.txt
done; done
for v in ctgn ctgn_nu; do CTGN_LIB_PATH=$PWD/ct_icp_amd/lib$v.so timeout 600 python scripts/iter_times.py B2 0 > $O/iter_B2_$v.txt 2> $O/err; echo "B2 $v:"; cat $O/iter_B2_$v.txt; done
for v in ctgn ctgn_nu; do CTGN_LIB_PATH=$PWD/ct_icp_amd/lib$v.so timeout 600 python scripts/iter_times.py C 0 > $O/iter_C_$v.txt 2> $O/err; echo "C $v:"; cat $O/iter_C_$v.txt; done
