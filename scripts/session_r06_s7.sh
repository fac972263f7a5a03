cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R="$PWD"; O=gpurun_out/s7; mkdir -p $O
for w in B2 D; do
  rm -rf gpurun_out/isa_tmp
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d "$R/gpurun_out/isa_tmp" -o pmc -- python "$R/scripts/isa_budget.py" run $w) > $O/run_$w.log 2>&1
  grep '^{"workload"' $O/run_$w.log > $O/schedule_$w.json
  f=$(find gpurun_out/isa_tmp -name "*counter_collection.csv" | head -1)
  python scripts/isa_budget.py report $O/schedule_$w.json $f > $O/isa_budget_$w.txt 2>&1
  cat $O/isa_budget_$w.txt
  tail -3 $O/run_$w.log | cut -c1-300
  rm -rf gpurun_out/isa_tmp
done
