"""Per-kernel averages of the counter CSVs tools/ssim_pmc.sh wrote (k_ssim launches on the 4K planes only)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for libdir in sorted(d for d in glob.glob(os.path.join(root, "*")) if os.path.isdir(d)):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(libdir, "set*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_ssim" not in r["Kernel_Name"]:
                continue
            if int(r["Grid_Size"]) < 64 * 1000:          # the small planes of the value check
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(libdir))
    for k in sorted(acc):
        v = acc[k]
        print(f"   {k:28s} {sum(v) / len(v):16.1f}   (n={len(v)})")
