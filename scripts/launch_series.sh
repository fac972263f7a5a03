#!/bin/bash
# per-launch duration of the search kernel over one solve (rocprofv3 kernel trace of the timed loop): is the first part of a solve slower?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R="$PWD"; rm -rf gpurun_out/series
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/series" -o t -- python "$R/bench.py" --steps ${1:-50} --warmup 0 --inner ${2:-}) > gpurun_out/series.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/series/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_accumulate_rows" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("search kernel us per launch:", " ".join(f"{x:.0f}" for x in d))
PY
