#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_7; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "config_e or frame or golden or sequence" > $O/pytest_fused.log 2>&1; echo "rc=$?" >> $O/pytest_fused.log
CTGN_FRAME_UNFUSED=1 timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "config_e or frame or golden or sequence" > $O/pytest_unfused.log 2>&1; echo "rc=$?" >> $O/pytest_unfused.log
timeout 300 python scripts/frame_prof.py > /dev/null 2> $O/frame_fused.txt
grep -v "^  File" $O/pytest_fused.log | tail -15 | cut -c1-300; echo ---; tail -4 $O/pytest_unfused.log; grep "python-side ctgn_frame" $O/frame_fused.txt; grep "frame_register us" $O/frame_fused.txt | tail -3
