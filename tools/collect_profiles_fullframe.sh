#!/bin/bash
# Run ON the GPU box (through gpurun): rocprofv3 kernel-trace stats of the full-frame (reference semantics) workload,
# 8 planes per step and 1 plane per step; summaries are copied to profiles/<tag>_fullframe_*.
#   gpurun -- 'bash tools/collect_profiles_fullframe.sh r02_q'
set -o pipefail
TAG=${1:-prof_ff}
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
for F in 8 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats$F" -- python3 bench.py --mode fullframe --steps 2 --cpu-frames 0 --ff-frames $F > "$OUT/stats$F.json" 2> "$OUT/stats$F.err" || { tail "$OUT/stats$F.err"; exit 1; }
  python3 bench.py --mode fullframe --steps 3 --ff-frames $F > "$OUT/bench$F.json" 2> "$OUT/bench$F.err" || { tail "$OUT/bench$F.err"; exit 1; }
  cat "$OUT/bench$F.json"
done
