#!/bin/bash
# Run ON the GPU box (through gpurun): SQ counters of k_ssim for one or more library builds, each counter set in its own
# rocprofv3 pass (kernel-trace + pmc only).   gpurun -- 'bash tools/ssim_pmc.sh TAG lib1.so lib2.so ...'
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_IFETCH SQ_ACTIVE_INST_VMEM"
 "GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"
 "SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_IFETCH_LEVEL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_CYCLES"
)
for lib in "$@"; do
  name=$(basename "$lib" .so)
  i=0
  for set in "${SETS[@]}"; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$name/set$i" -- python3 tools/ab_ssim.py "$lib" --rounds 1 --reps 2 > "$OUT/$name.set$i.log" 2> "$OUT/$name.set$i.err" || { tail -5 "$OUT/$name.set$i.err"; }
    i=$((i+1))
  done
done
python3 tools/ssim_pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
