#!/bin/bash
# round 4, GPU session 6: fused map update inside ctgn_frame. Parity suite, frame-pipeline phase marks fused / unfused, bench extras.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_6; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/frame_prof.py > /dev/null 2> $O/frame_fused.txt
CTGN_FRAME_UNFUSED=1 timeout 300 python scripts/frame_prof.py > /dev/null 2> $O/frame_unfused.txt
timeout 900 python bench.py --sub none --no-pmc --steps 100 > $O/bench_b2.json 2> $O/bench_b2.err; cp bench_detail.json $O/bench_b2_detail.json
tail -n 8 $O/pytest_gpu.log; grep "ctgn_frame\|frame_update_map\|want_all=True" $O/frame_fused.txt | tail -12; echo ---; grep "ctgn_frame\|frame_update_map" $O/frame_unfused.txt | tail -8; cat $O/bench_b2.json | cut -c1-2500
