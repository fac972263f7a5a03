#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
REPS=2 bash scripts/ab3.sh > gpurun_out/ab3.txt 2>&1; cat gpurun_out/ab3.txt
