for sb in 6 4 2; do
  echo "hier sb=$sb"; WM_RF_HIER_SB=$sb python3 bench.py --mode fullframe --steps 3 --cpu-frames 0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['embed_ms_per_plane'], d['roofline']['note'][:20])"
done
echo flat; WM_RF_HIER=0 python3 bench.py --mode fullframe --steps 3 --cpu-frames 0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['embed_ms_per_plane'])"
for q in 1 2 4; do echo "hier queues=$q"; WM_RF_QUEUES=$q python3 bench.py --mode fullframe --steps 3 --cpu-frames 0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['embed_ms_per_plane'])"; done
