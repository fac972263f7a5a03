#!/bin/bash
# session 28: the N = 2 path of bench.py at FULL size (workload D, 462 k keypoints per rank: the split pool check is on in sharded mode),
# two gloo ranks on the one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_28; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29579 bench.py --gpus 2 --dist-backend gloo --steps 10 --warmup 2 --no-pmc > $O/n2_full.txt 2> $O/n2_full.err ) 2> $O/time.txt
grep '^{' $O/n2_full.txt | tail -1 | cut -c1-2500; tail -3 $O/n2_full.err; cat $O/time.txt
