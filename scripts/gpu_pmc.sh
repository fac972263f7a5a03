#!/bin/bash
# PMC passes (separate runs, --pmc only with --kernel-trace) for the accumulate kernel. Output: gpurun_out/pmc_*.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
R="$PWD"
[ -f gpurun_out/counters.txt ] || (rocprofv3 -L > gpurun_out/counters.txt 2>&1)
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rm -rf gpurun_out/pmc$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $line --output-format csv -d "$R/gpurun_out/pmc$i" -o pmc -- python "$R/bench.py" --steps 200 --warmup 0 --inner ${BENCH_ARGS:-}) > gpurun_out/pmc$i.log 2>&1
  f=$(find gpurun_out/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > gpurun_out/pmc_$i.txt; cat gpurun_out/pmc_$i.txt; else tail -5 gpurun_out/pmc$i.log; fi
  find gpurun_out/pmc$i -type f -size +1M -delete
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
LIST
