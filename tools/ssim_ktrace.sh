#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel-trace durations of k_ssim / k_sum_f64 per library build.
#   gpurun -- 'bash tools/ssim_ktrace.sh TAG lib1.so lib2.so ...'
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
for lib in "$@"; do
  name=$(basename "$lib" .so)
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/ab_ssim.py "$lib" --rounds 2 --reps 3 > "$OUT/$name.log" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; }
  python3 - "$OUT/$name" "$name" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "k_ssim" if "k_ssim" in n else ("k_sum_f64" if "k_sum_f64" in n else None)
        if key is None: continue
        if key == "k_ssim" and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0)) * int(r.get("Grid_Size_Y", 1) or 1) < 64 * 1000: continue
        acc[key + ("<u8,u8>" if "IhhL" in n else "<u8,f32>" if "IhfL" in n else "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    print(f"{sys.argv[2]:18s} {k:18s} n={len(v):4d} median {v[len(v)//2]:8.2f} us  min {v[0]:8.2f}")
PY
done | tee "$OUT/summary.txt"
