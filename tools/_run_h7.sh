for HW in 4 8; do for H in 1 0; do for q in 3 4 6 8; do
  echo -n "GPU_MAX_HW_QUEUES=$HW hier=$H queues=$q: "
  GPU_MAX_HW_QUEUES=$HW WM_RF_HIER=$H WM_RF_QUEUES=$q python3 bench.py --mode fullframe --steps 3 --cpu-frames 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['embed_ms_per_plane'],3))"
done; done; done
