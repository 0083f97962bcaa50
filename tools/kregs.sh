#!/bin/bash
# Device-only compile of csrc/<file>.hip to assembly and a table of VGPRs / spills / scratch per kernel.
#   tools/kregs.sh wmhip [filter-regex]
F=${1:-wmhip}; PAT=${2:-.}
SRC="$(dirname "$0")/../digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd/csrc/$F.hip"
OUT=/tmp/kregs_$F.s
hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S -o "$OUT" "$SRC" 2>&1 | grep -E "error" -A4
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|name):" "$OUT" | paste - - - - \
  | sed -E 's/\s+\.name:\s+//; s/\.private_segment_fixed_size:/scratch/; s/\.vgpr_count:/vgpr/; s/\.vgpr_spill_count:/spill/' | grep -E "$PAT"
