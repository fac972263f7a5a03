cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R="$PWD"; O=gpurun_out/s8; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
for w in B2 D; do for rep in 1 2; do timeout 600 python scripts/iter_times.py $w 0 2>&1 | grep '^{' >> $O/iter_times.txt; done; done; cat $O/iter_times.txt
for w in B2 D; do
  rm -rf gpurun_out/isa_tmp
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d "$R/gpurun_out/isa_tmp" -o pmc -- python "$R/scripts/isa_budget.py" run $w) > $O/run_$w.log 2>&1
  grep '^{"workload"' $O/run_$w.log > $O/schedule_$w.json
  f=$(find gpurun_out/isa_tmp -name "*counter_collection.csv" | head -1)
  python scripts/isa_budget.py report $O/schedule_$w.json $f > $O/isa_budget_$w.txt 2>&1
  cat $O/isa_budget_$w.txt
  rm -rf gpurun_out/isa_tmp
done
