#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_18; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "frame or sequence or config_e or golden" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python scripts/frame_prof.py > /dev/null 2> $O/frame_marks.txt
timeout 1500 python scripts/sequence_run.py --config-e --scale 10 --solver GN,CERES --out $O/config_e_n1.json > /dev/null 2> $O/config_e.err
grep -v "^  File" $O/pytest_gpu.log | tail -n 5 | cut -c1-300; grep "ctgn_frame\|frame_register us" $O/frame_marks.txt | tail -5; tail -24 $O/config_e.err
