timeout -k 10 600 python -m pytest tests/test_gpu_fullframe_twolevel.py -x -q 2>&1 | tail -2
for F in 64 16; do echo -n "planes=$F: "; python3 bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['embed_ms_per_plane'],3))"; done
