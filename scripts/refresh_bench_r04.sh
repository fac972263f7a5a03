#!/bin/bash
# The bench-line part of scripts/refresh_profiles_r04.sh alone (after a change of bench.py that leaves the kernels as they are).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
P=gpurun_out/profiles_r04; rm -rf $P; mkdir -p $P
( time timeout 1500 python bench.py --steps 20 --warmup 5 > $P/r04_bench_n1.json 2> gpurun_out/bench_default.err ) 2> gpurun_out/bench_default_time.txt
cp bench_detail.json $P/r04_bench_detail.json; grep real gpurun_out/bench_default_time.txt > $P/r04_bench_n1_wallclock.txt
timeout 1500 python bench.py --sub none --config-e-scale 0 > $P/r04_bench_n1_200steps.json 2> /dev/null; cp bench_detail.json $P/r04_bench_200steps_detail.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --dist-backend gloo --steps 10 --warmup 2 --clock-warm 5 --d-sweeps 2 --d-radius 60 --no-pmc 2>/dev/null | grep '^{' | tail -1 > $P/r04_bench_n2_gloo_rehearsal.json
ls -la $P; cat $P/r04_bench_n1.json | cut -c1-400; cat $P/r04_bench_n1_wallclock.txt; tail -3 gpurun_out/bench_default.err
