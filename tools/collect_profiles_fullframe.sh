#!/bin/bash
# Run ON the GPU box (through gpurun): the full-frame (reference semantics) workload under rocprofv3 -
# kernel-trace stats, then FETCH_SIZE, WRITE_SIZE and SQ counters in passes of their own (never combined
# with other trace domains) - at the bench's default batch and at one plane per step.  Raw CSVs stay under
# gpurun_out/<tag>/; tools/summarize_profiles_fullframe.py <tag> turns them into profiles/<tag>_fullframe_*.
#   gpurun -- 'bash tools/collect_profiles_fullframe.sh r04_a'
set -o pipefail
TAG=${1:-prof_ff}
shift
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
FDEF=$(python3 -c "import bench,sys; sys.argv=['bench.py']; print(bench.parse().ff_frames)")
for F in ${FF_LIST:-$FDEF 1}; do
  ARGS="bench.py --mode fullframe --steps 2 --cpu-frames 0 --ff-frames $F $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats$F" -- python3 $ARGS > "$OUT/stats$F.json" 2> "$OUT/stats$F.err" || { tail "$OUT/stats$F.err"; exit 1; }
  echo "stats $F done"
  for P in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "lds:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    N=${P%%:*}; C=${P#*:}
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$N$F" -- python3 $ARGS > "$OUT/$N$F.json" 2> "$OUT/$N$F.err" || { echo "pmc pass $N failed"; tail -3 "$OUT/$N$F.err"; }
    echo "pmc $N $F done"
  done
  python3 bench.py --mode fullframe --steps 3 --ff-frames $F --cpu-frames 0 $* > "$OUT/bench$F.json" 2> "$OUT/bench$F.err" || { tail "$OUT/bench$F.err"; exit 1; }
  cat "$OUT/bench$F.json"
  python3 tools/summarize_profiles_fullframe.py "$OUT" $F
  cp "$OUT"/stats$F/*/*kernel_stats.csv "$OUT/kernel_stats$F.csv"
  # the raw per-dispatch CSVs are tens of MB: only the summaries travel back
  rm -rf "$OUT"/stats$F "$OUT"/fetch$F "$OUT"/write$F "$OUT"/sq$F "$OUT"/lds$F
done
