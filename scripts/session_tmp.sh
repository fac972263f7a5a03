cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s14; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -k "not rehearsal" > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench.err ) 2> $O/bench_time.txt
cp bench_detail.json $O/bench_detail.json; cut -c1-3500 $O/bench_n1.json; tail -2 $O/bench.err; cat $O/bench_time.txt
