#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02r; mkdir -p $O
for fs in 0 2 40; do
  WM_RF_FULL_SWEEPS=$fs python bench.py --mode fullframe --steps 2 --cpu-frames 0 --ff-frames 1 > $O/fs$fs.json 2> $O/fs$fs.err || { tail $O/fs$fs.err; exit 1; }
  python -c "import json; j=json.load(open('$O/fs$fs.json')); print('full_sweeps $fs:', round(j['embed_ms_per_plane'],2), 'ms', j['roofline']['note'][:30])"
done
