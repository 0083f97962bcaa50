cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for D in 0 1 2 3 16 32 48 64 112; do
  rm -rf gpurun_out/dbg$D
  WM_RF_HDBG=$D WM_RF_QUEUES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dbg$D -- python3 bench.py --mode fullframe --steps 1 --cpu-frames 0 --ff-frames 5 > /dev/null 2>&1
  echo "dbg $D"; python3 tools/kstats.py gpurun_out/dbg$D | grep "k_hgram\|k_happly"
  rm -rf gpurun_out/dbg$D
done
