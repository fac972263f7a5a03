cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
MODE=iter WORKLOADS="B2 D" REPS=3 scripts/ab.sh prev:.ab/libctgn_prev.so cur > $O/ab_compact_vs_prev.txt 2>&1; cat $O/ab_compact_vs_prev.txt
