# A/B of the neighbourhood sums handed over by the pool checks (CTGN_POOL_SUMS): parity first, then per-iteration search-kernel times and
# step times of fresh solves on B2 and D with (a) this build, (b) this build with tuning pool_sums=0, (c) a build without the code.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $O/pytest_parity.log 2>&1; echo "rc=$?" >> $O/pytest_parity.log
tail -5 $O/pytest_parity.log
for w in B2 D; do
  for rep in 1 2; do
    for v in sums off nosums; do
      lib="$PWD/ct_icp_amd/libctgn.so"; tun=""
      [ $v = off ] && tun="pool_sums=0"
      [ $v = nosums ] && lib="$PWD/.ab/libctgn_nosums.so"
      echo "== $w $v rep $rep" >> $O/iter_times.txt
      CTGN_TUNING="$tun" CTGN_LIB_PATH=$lib timeout 600 python scripts/iter_times.py $w 0 2>&1 | grep '^{' >> $O/iter_times.txt
    done
  done
done
cat $O/iter_times.txt
for w in B2 D; do timeout 600 python scripts/rows_prof3.py $w 4 2>&1 | grep '^{' | cut -c1-400 >> $O/rows_prof3.txt; done
cat $O/rows_prof3.txt
# per-kernel times of the timed loop, with and without
for w in B2 D; do
  for v in sums nosums; do
    lib="$PWD/ct_icp_amd/libctgn.so"; [ $v = nosums ] && lib="$PWD/.ab/libctgn_nosums.so"
    st=200; [ $w = D ] && st=20
    rm -rf gpurun_out/prof_tmp
    (cd /tmp && CTGN_LIB_PATH=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_tmp" -o trace -- python "$OLDPWD/bench.py" --workload $w --steps $st --warmup 0 --inner) > /dev/null 2>&1
    echo "== $w $v" >> $O/kernel_stats.txt
    find gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1 | xargs -r cat | cut -d, -f1-5 | head -8 | cut -c1-220 >> $O/kernel_stats.txt
    rm -rf gpurun_out/prof_tmp
  done
done
cat $O/kernel_stats.txt
