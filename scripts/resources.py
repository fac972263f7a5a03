#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel of libctgn (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
Usage: python scripts/resources.py [> profiles/rNN_kernel_resources.txt]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "ct_icp_amd", "csrc")
rows = []
for src in ("ctgn_api.hip", "ctgn_devmap.hip"):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                          "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src], cwd=csrc, capture_output=True, text=True).stderr
    cur = None
    for ln in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", ln)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':84s} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch B/lane':>15} {'waves/SIMD':>11} {'LDS B/block':>12}")
for r, name in zip(rows, names):
    name = re.sub(r"\((ctgn::|double|unsigned|int|float|char|void|const|long|uint).*", "", name).replace("void ", "").replace("ctgn::", "")
    print(f"{name[:84]:84s} {r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('TotalSGPRs', r.get('SGPRs', '?')):>5} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>15} {r.get('Occupancy [waves/SIMD]', '?'):>11} {r.get('LDS Size [bytes/block]', '?'):>12}")
