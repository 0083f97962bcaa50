bash tools/_run_h5.sh | head -16
for q in 2 3; do echo "hier sb=6 queues=$q"; WM_RF_QUEUES=$q python3 bench.py --mode fullframe --steps 3 --cpu-frames 0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['embed_ms_per_plane'])"; done
FF_LIST=16 bash tools/collect_profiles_fullframe.sh r04_c | grep -v "^{" | tail -7
