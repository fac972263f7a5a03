#!/bin/bash
# round 4, GPU session 10: wave-shared probes of the 125-voxel sweep (bit 22 = off). Parity suite; D and C per-iteration search times.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_10; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 900 python scripts/iter_times.py D 0 4194304 0 4194304 > $O/iter_D.txt 2> $O/iter_D.err
timeout 600 python scripts/iter_times.py C 0 4194304 > $O/iter_C.txt 2> $O/iter_C.err
grep -v "^  File" $O/pytest_gpu.log | tail -n 12 | cut -c1-300; cat $O/iter_D.txt $O/iter_C.txt
