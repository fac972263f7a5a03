#!/bin/bash
# session 26: config E (KITTI lengths / 100 and / 10, GN route) with the scans in page-locked memory against pageable memory
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_26; mkdir -p $O
export TMPDIR=/tmp
for mode in "" "--pageable" "" "--pageable"; do
  timeout 600 python scripts/sequence_run.py --config-e --scale 100 $mode 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scale 100', d['scan_arrays'], round(d['frames_per_sec'],1), 'frames/s, failures', d['failures'], 'err max', round(d['err_tr_max'],4))"
done
for mode in "" "--pageable"; do
  timeout 900 python scripts/sequence_run.py --config-e --scale 10 $mode --out $O/e10$mode.json 2> $O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scale 10', d['scan_arrays'], round(d['frames_per_sec'],1), 'frames/s, failures', d['failures'], 'err max', round(d['err_tr_max'],4))"
done
tail -3 $O/err.txt
