timeout -k 10 200 python3 tools/hier_check.py 2>&1 | tail -3
for F in 16 48 64 96; do for q in 3; do echo -n "hier planes=$F queues=$q: "; WM_RF_QUEUES=$q python3 bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['embed_ms_per_plane'],3))"; done; done
FF_LIST=16 bash tools/collect_profiles_fullframe.sh r04_c | grep -v "^{" | tail -8
