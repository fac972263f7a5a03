cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s13; mkdir -p $O; R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wave_shared_probes or config_d_ouster or config_c_nclt" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
WORKLOADS="D" REPS=2 STEPS=30 WARM=30 timeout 2400 scripts/ab.sh cur "stage0::stage_lds=1" "stage8::stage_lds=1,tile_chunk=8" "stage16::stage_lds=1,tile_chunk=16" "chunk8::tile_chunk=8" > $O/ab_D.txt 2>&1
cat $O/ab_D.txt
for c in 0 8; do timeout 600 python scripts/stage_probe.py D $c 2>&1 | grep '^{' | head -2; done > $O/stage_probe_D.txt 2>&1; cat $O/stage_probe_D.txt
i=0
for tun in "" "stage_lds=1,tile_chunk=8"; do
 for line in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf $O/pmc$i
  (cd /tmp && CTGN_TUNING="$tun" timeout 400 rocprofv3 --kernel-trace --pmc $line --output-format csv -d "$R/$O/pmc$i" -o pmc -- python "$R/bench.py" --workload D --steps 10 --warmup 0 --inner --no-cpu-baseline --no-pmc --no-extras --sub none) > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*counter_collection.csv" | head -1)
  echo "== tuning '$tun'" > $O/pmc_$i.txt
  if [ -n "$f" ]; then python scripts/pmc_summary.py "$f" >> $O/pmc_$i.txt; else tail -5 $O/pmc$i.log >> $O/pmc_$i.txt; fi
  find $O/pmc$i -type f -size +1M -delete
 done
done
cat $O/pmc_*.txt | head -120
