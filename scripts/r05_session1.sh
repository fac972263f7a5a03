#!/bin/bash
# Round-5 session 1 on one box: parity tests of the new build, then A/B of (a) the per-XCD pre-sums (B2), (b) the 125-voxel sweep's early
# batch exit + table reuse + chunked tiles (D), (c) the zero-copy end of small-frame solves. Logs under gpurun_out/s1/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/s1; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extras --sub none --config-e-scale 0"
line() { python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step %.4f kernel_ms %.4f first %.4f later %.4f parity %s" % (d["ms_per_step"], r["kernel_ms_avg"], r["first_iteration"]["kernel_ms"], r["later_iterations"]["kernel_ms"], d.get("parity_m_rad")))'; }
base=$PWD/ct_icp_amd/libctgn_base.so
for rep in 1 2; do
  echo "B2 base: $(CTGN_LIB_PATH=$base timeout 300 python bench.py $B 2>>$O/err.log | line)"
  echo "B2 cur : $(timeout 300 python bench.py $B 2>>$O/err.log | line)"
  echo "B2 cur xcd_reduce=0: $(CTGN_TUNING=xcd_reduce=0 timeout 300 python bench.py $B 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_B2.txt
for rep in 1 2; do
  echo "D base: $(CTGN_LIB_PATH=$base timeout 400 python bench.py --workload D $B 2>>$O/err.log | line)"
  echo "D cur : $(timeout 400 python bench.py --workload D $B 2>>$O/err.log | line)"
  echo "D cur chunk16: $(CTGN_TUNING=tile_chunk=16 timeout 400 python bench.py --workload D $B 2>>$O/err.log | line)"
  [ $rep = 1 ] && echo "D cur chunk4: $(CTGN_TUNING=tile_chunk=4 timeout 400 python bench.py --workload D $B 2>>$O/err.log | line)"
done 2>&1 | tee $O/ab_D.txt
timeout 300 python scripts/register_time.py B1 zero_copy_end=0,1 2>>$O/err.log | tee $O/ab_register.txt
tail -5 $O/err.log
