#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_23; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "frame or sequence or config_e or golden or sort or device_resident" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/frame_prof.py > /dev/null 2> $O/frame_ahead.txt
CTGN_FRAME_SYNC=1 timeout 300 python scripts/frame_prof.py > /dev/null 2> $O/frame_sync.txt
grep -v "^  File" $O/pytest_gpu.log | tail -n 12 | cut -c1-300; echo "--- sized ahead"; grep "python-side\|frame_register us" $O/frame_ahead.txt | tail -9; echo "--- waits for counts"; grep "python-side ctgn_frame\|frame_register us" $O/frame_sync.txt | tail -5
