#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02n; mkdir -p $O
for q in 1 2 3 4; do for at in 1 2 4; do
  WM_RF_QUEUES=$q WM_RF_APPLY_TILES=$at python bench.py --mode fullframe --steps 3 --cpu-frames 0 > $O/q$q.a$at.json 2> $O/q$q.a$at.err || { tail $O/q$q.a$at.err; exit 1; }
  python -c "import json; j=json.load(open('$O/q$q.a$at.json')); print('queues $q apply_tiles $at:', round(j['value'],1), 'fps, embed ms/plane', round(j['embed_ms_per_plane'],2))"
done; done
