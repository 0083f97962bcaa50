#!/bin/bash
# Run ON the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes of
# tools/pixel_roofline.py (the HBM-bound kernels either side of the hot path); tools/summarize_profiles_pixel.py turns the
# raw CSVs into profiles/<tag>_pixel_*.
#   gpurun -- 'bash tools/collect_profiles_pixel.sh r03_p'
set -o pipefail
TAG=${1:-prof_px}
OUT=gpurun_out/$TAG
CMD="python3 tools/pixel_roofline.py --frames 8 --reps 10"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
$CMD --json "$OUT/events.json" > "$OUT/events.log" 2> "$OUT/events.err" || { tail "$OUT/events.err"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2> "$OUT/stats.err" || { tail "$OUT/stats.err"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2> "$OUT/fetch.err" || { tail "$OUT/fetch.err"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2> "$OUT/write.err" || { tail "$OUT/write.err"; exit 1; }
cat "$OUT/events.log"
