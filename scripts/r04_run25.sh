#!/bin/bash
# session 25: page-locked scan arrays read / written in place by ctgn_frame_register (no staging): the new GPU test, the frame tests,
# then the frame-pipeline part of bench.py (pageable vs page-locked) with the C-side marks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_25; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "page_locked or frame" 2>&1 | tail -5
CTGN_FRAME_TIMING=1 timeout 900 python - > $O/frames.txt 2> $O/marks.txt <<'P'
import json, sys
sys.path.insert(0, ".")
import bench
import ct_icp_amd as cia
from ct_icp_amd import se3, synthetic as syn
inp = bench.make_inputs(0, 20)
fs = bench.measure_frame_stages(cia, inp, syn, se3, 0)
print(json.dumps(fs["frame_pipeline"]))
P
echo "---"; tail -1 $O/frames.txt; tail -8 $O/marks.txt
