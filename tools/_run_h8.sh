for F in 16 32 48 64; do for H in 1 0; do for q in 2 3; do
  echo -n "planes=$F hier=$H queues=$q: "
  WM_RF_HIER=$H WM_RF_QUEUES=$q python3 bench.py --mode fullframe --steps 2 --cpu-frames 0 --ff-frames $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['embed_ms_per_plane'],3))"
done; done; done
