#!/bin/bash
# A/B of bench.py configurations on one box: prints ms_per_step, search-kernel ms and keypoints/s per configuration. Usage: scripts/ab_bench.sh "<args A>" "<args B>" ...
for cfg in "$@"; do
  out=$(python bench.py --steps 50 --warmup 10 --no-pmc --no-cpu-baseline --no-extras $cfg 2>/dev/null | tail -1)
  echo "$cfg => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms/step %.4f  kernel_ms %.4f  kp/s %.3e  frac %.3f  parity %s" % (d["ms_per_step"], r["kernel_ms_avg"], d["value"], r["frac"], d.get("parity_m_rad")))')"
done
