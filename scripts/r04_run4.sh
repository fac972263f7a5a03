#!/bin/bash
# round 4, GPU session 4: counters of k_pool_check (split launches forced on the B2 sweep): what is the pool check bound by?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_4; mkdir -p $O
export TMPDIR=/tmp
R="$PWD"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  (cd /tmp && CTGN_SPLIT=1 timeout 300 rocprofv3 --kernel-trace --pmc $line --output-format csv -d "$R/$O/pmc$i" -o pmc -- python "$R/bench.py" --steps 60 --warmup 0 --inner) > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" > $O/pmc_$i.txt; cat $O/pmc_$i.txt; else tail -5 $O/pmc$i.log; fi
  rm -rf $O/pmc$i
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
LIST
