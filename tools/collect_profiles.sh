#!/bin/bash
# Run ON the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC
# passes of the bench command; raw CSVs land in gpurun_out/<tag>/, the judged
# summaries are produced by tools/summarize_profiles.py into profiles/.
#   gpurun -- 'bash tools/collect_profiles.sh r01_c'
set -o pipefail
TAG=${1:-prof}
OUT=gpurun_out/$TAG
CMD="python3 bench.py --steps 5 --warmup 2 --cpu-frames 0 --quick"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.json" 2> "$OUT/stats.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/sq" -- $CMD > "$OUT/sq.json" 2> "$OUT/sq.err" || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.json" 2> "$OUT/fetch.err" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.json" 2> "$OUT/write.err" || exit 1
python bench.py --steps 20 --warmup 3 > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err" || exit 1
cat "$OUT/bench_unprofiled.json"
