#!/bin/bash
# Round-5 evidence that the main refresh left out: config C's PMC passes, the two-rank (gloo) rehearsal of bench.py's N > 1 path on one GPU,
# the frame pipeline's host-clock marks. Results under gpurun_out/extras/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; P=gpurun_out/extras; rm -rf $P; mkdir -p $P
rm -f gpurun_out/pmc_?.txt
BENCH_ARGS="--workload C" scripts/gpu_pmc.sh > gpurun_out/pmc_all_C.log 2>&1
for i in 1 2 3 4 5; do [ -f gpurun_out/pmc_$i.txt ] && cp gpurun_out/pmc_$i.txt $P/workloadC_r05_pmc_pass$i.txt; done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --dist-backend gloo --steps 10 --warmup 2 --clock-warm 5 --d-sweeps 2 --d-radius 60 --no-pmc 2>/dev/null | grep '^{' | tail -1 > $P/r05_bench_n2_gloo_rehearsal.json
timeout 300 python scripts/frame_prof.py 2>&1 | grep -v amdgpu.ids | tail -16 > $P/r05_frame_pipeline_marks.txt
find gpurun_out/pmc? -type f -size +1M -delete 2>/dev/null
ls -la $P; cut -c1-600 $P/r05_bench_n2_gloo_rehearsal.json; tail -6 $P/r05_frame_pipeline_marks.txt
