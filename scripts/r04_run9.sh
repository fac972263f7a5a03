#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_9; mkdir -p $O
export TMPDIR=/tmp
MASTER_ADDR=127.0.0.1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29567 tests/config_e_worker.py > $O/worker.out 2> $O/worker.err; echo "worker rc=$?"
grep -v "^W\|^$" $O/worker.err | grep -B2 -A12 "Traceback" | head -40; cat $O/worker.out | tail -2
timeout 900 python scripts/sequence_run.py --config-e --scale 10 --only 0 --per-frame --out $O/e0_fused.json > /dev/null 2> $O/e0_fused.err
CTGN_FRAME_UNFUSED=1 timeout 900 python scripts/sequence_run.py --config-e --scale 10 --only 0 --per-frame --out $O/e0_unfused.json > /dev/null 2> $O/e0_unfused.err
timeout 900 python scripts/sequence_run.py --config-e --scale 10 --only 0 --per-frame --solver CERES --out $O/e0_ceres.json > /dev/null 2> $O/e0_ceres.err
tail -1 $O/e0_fused.err; tail -1 $O/e0_unfused.err; tail -1 $O/e0_ceres.err
