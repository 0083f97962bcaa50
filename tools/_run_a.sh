cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_fullframe_twolevel.py -x -q 2>&1 | tail -2
for F in 64 16; do echo -n "planes=$F: "; python3 bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['embed_ms_per_plane'],3))"; done
export WM_RF_QUEUES=1
for D in 0 64 384; do
  rm -rf /tmp/hp_$D
  WM_RF_HDBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp_$D -- python3 bench.py --mode fullframe --steps 1 --warmup 0 --cpu-frames 0 --ff-frames 21 > /tmp/hp_$D.json 2> /tmp/hp_$D.err
  echo "dbg=$D: $(python3 tools/kstats.py /tmp/hp_$D | grep -E 'k_happly_h|k_rf_inner|k_hgram_h' | tr '\n' '|')"
done
