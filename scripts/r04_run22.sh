#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_22; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "ties or accumulate or km_scale or robust or lattice" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for rep in 1 2; do for v in ctgn_inl ctgn; do
  CTGN_LIB_PATH=$PWD/ct_icp_amd/lib$v.so timeout 600 python scripts/iter_times.py B2 0 > $O/iter_B2_${v}_$rep.txt 2> $O/err; echo "B2 $v:"; cat $O/iter_B2_${v}_$rep.txt
done; done
for v in ctgn_inl ctgn; do CTGN_LIB_PATH=$PWD/ct_icp_amd/lib$v.so timeout 600 python scripts/iter_times.py B1 0 > $O/iter_B1_$v.txt 2> $O/err; echo "B1 $v:"; cat $O/iter_B1_$v.txt; done
for v in ctgn_inl ctgn; do CTGN_LIB_PATH=$PWD/ct_icp_amd/lib$v.so timeout 600 python scripts/iter_times.py D 0 > $O/iter_D_$v.txt 2> $O/err; echo "D $v:"; cat $O/iter_D_$v.txt; done
