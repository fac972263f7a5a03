import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "k_accumulate" not in k and "k_reduce" not in k and "k_residual" not in k and "k_pool_check" not in k:
        continue
    agg[k.split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
