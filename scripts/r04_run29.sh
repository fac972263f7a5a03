#!/bin/bash
# session 29: the solve loops stop enqueuing once the stop test has fired (progress word two iterations behind): whole GPU suite, then the
# Register / frame numbers with and without it (CTGN_NO_EARLY_STOP=1), config E / 100 both ways
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_29; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x 2>&1 | tail -3
for env in "" "CTGN_NO_EARLY_STOP=1" "" "CTGN_NO_EARLY_STOP=1"; do
  env $env timeout 600 python bench.py --steps 10 --warmup 3 --no-pmc --sub none --config-e-scale 100 --no-cpu-baseline > $O/line.txt 2> $O/err.txt
  python - "$env" <<'P'
import json, sys
l = json.loads(open("gpurun_out/r04_29/line.txt").read().strip().splitlines()[-1])
print(sys.argv[1] or "early stop on", "| Register", l["frames_per_sec"], "| robust", l["robust_route"], "| frame", l["frame_pipeline"], "| E/100", l["config_e"]["frames_per_sec"], "| B2", l["ms_per_step"])
P
done
tail -2 $O/err.txt
