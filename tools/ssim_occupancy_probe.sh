#!/bin/bash
# Run ON the GPU box: k_ssim kernel-trace durations against the grid size (workgroups = ceil(W/64) x ceil(H/34), one wave each):
# flat time from 1 to 4 waves per SIMD = a latency chain per row; time growing with the waves = a throughput limit.
#   gpurun -- 'bash tools/ssim_occupancy_probe.sh TAG lib.so'
set -o pipefail
TAG=$1; LIB=$2
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
for hw in "544 1024" "1088 2048" "1088 4096" "1632 4096" "2176 4096" "2176 8192"; do
  set -- $hw
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/h$1w$2" -- python3 tools/ab_ssim.py "$LIB" --H $1 --W $2 --frames 2 --rounds 2 --reps 3 > "$OUT/h$1w$2.log" 2> "$OUT/h$1w$2.err" || { tail -3 "$OUT/h$1w$2.err"; }
  python3 - "$OUT/h$1w$2" $1 $2 <<'PY'
import csv, glob, sys, os
H, W = int(sys.argv[2]), int(sys.argv[3])
v = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ssim<unsigned char, unsigned char" in r["Kernel_Name"] or ("k_ssim" in r["Kernel_Name"] and "IhhL" in r["Kernel_Name"]):
            g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * int(r.get("Grid_Size_Y", 1) or 1)
            if g >= 64 * 200:
                v.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
v.sort()
wg = ((W + 63) // 64) * ((H + 33) // 34)
print(f"H {H:5d} W {W:5d}  workgroups {wg:6d} = {wg / 1024:5.2f} waves per SIMD   k_ssim median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}   ({v[len(v)//2] / 44 * 1e3:6.0f} ns per input row of a wave)")
PY
done | tee "$OUT/summary.txt"
