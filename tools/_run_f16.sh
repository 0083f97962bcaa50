WM_RF_HIER_F16=3 timeout -k 10 300 python -m pytest tests/test_gpu_fullframe_twolevel.py -x -q 2>&1 | tail -6
WM_RF_HIER_F16=3 WM_RF_HIER=1 timeout -k 10 200 python3 tools/hier_check.py 2>&1 | tail -7
for F16 in 1 3; do for F in 16 48; do echo -n "f16=$F16 planes=$F: "; WM_RF_HIER_F16=$F16 WM_RF_HIER=1 python3 bench.py --mode fullframe --steps 3 --ff-frames $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['embed_ms_per_plane'],3), d.get('parity'))"; done; done
