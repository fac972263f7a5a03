cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s15; mkdir -p $O
WORKLOADS="B2" REPS=3 STEPS=100 WARM=100 timeout 2400 scripts/ab.sh cur lcap112:.ab/libctgn_lcap112.so lcap128:.ab/libctgn_lcap128.so > $O/ab.txt 2>&1
WORKLOADS="D" REPS=2 STEPS=30 WARM=30 timeout 2400 scripts/ab.sh cur lcap112:.ab/libctgn_lcap112.so >> $O/ab.txt 2>&1
cat $O/ab.txt
