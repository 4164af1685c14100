#!/usr/bin/env python3
"""Average per-dispatch PMC counter values per kernel from a rocprofv3 --pmc csv directory."""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
for f in files:
    for row in csv.DictReader(open(f)):
        m = re.search(r"((?:k|kb)_\w+(<[^>]*>)?)", row["Kernel_Name"])
        k = m.group(1) if m else "other"
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    if not k.startswith(("k_", "kb_")):
        continue
    print(k, " ".join(f"{c}={acc[k][c]/cnt[k][c]:.4g}" for c in sorted(acc[k])), f"(n={max(cnt[k].values())})")
