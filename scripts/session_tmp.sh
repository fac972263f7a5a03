cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s14; mkdir -p $O
WORKLOADS="B2" REPS=3 STEPS=100 WARM=100 timeout 2400 scripts/ab.sh cur lcap80:.ab/libctgn_lcap80.so wps4:.ab/libctgn_wps4.so > $O/ab.txt 2>&1
WORKLOADS="D" REPS=2 STEPS=30 WARM=30 timeout 2400 scripts/ab.sh cur lcap80:.ab/libctgn_lcap80.so wps4:.ab/libctgn_wps4.so >> $O/ab.txt 2>&1
cat $O/ab.txt
