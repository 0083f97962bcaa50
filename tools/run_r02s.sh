#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullframe.py tests/test_gpu_state_reuse.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for f in 8 1; do for cpw in 1 2 4 8; do for q in 1 2; do
  WM_RF_QUEUES=$q WM_RF_GRAM_CPW=$cpw python bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames $f > $O/f$f.c$cpw.q$q.json 2> $O/f$f.c$cpw.q$q.err || { tail $O/f$f.c$cpw.q$q.err; exit 1; }
  python -c "import json; j=json.load(open('$O/f$f.c$cpw.q$q.json')); print('planes $f cpw $cpw queues $q:', round(j['value'],1), 'fps, embed ms/plane', round(j['embed_ms_per_plane'],2))"
done; done; done
