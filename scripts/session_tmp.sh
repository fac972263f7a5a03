cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robust.py -m gpu -q --timeout 900 -p no:cacheprovider -x -k "not rehearsal" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
MODE=iter WORKLOADS="B2 B1" REPS=2 MASKS="0 1073741824" scripts/ab.sh cur > $O/ab_compact.txt 2>&1; cat $O/ab_compact.txt
