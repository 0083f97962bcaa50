#!/bin/bash
# inner-solve token between the two queues (WM_RF_INNER_CHAIN=1, new default) vs free-running queues (=0)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02ab; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullframe.py tests/test_gpu_state_reuse.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for c in 0 1 0 1; do for f in 3 8; do
  WM_RF_INNER_CHAIN=$c python bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames $f > $O/c$c.$f.json 2> $O/c$c.$f.err || { tail $O/c$c.$f.err; exit 1; }
  python -c "import json; j=json.load(open('$O/c$c.$f.json')); print('chain $c planes $f:', round(j['value'],1), 'fps, embed ms/plane', round(j['embed_ms_per_plane'],2))"
done; done
for c in 0 1; do
  WM_RF_INNER_CHAIN=$c rocprofv3 --kernel-trace --output-format csv -d $O/tl$c -- python3 tools/prof_ff_batch.py 8 > $O/tl$c.log 2>&1 || { tail $O/tl$c.log; exit 1; }
  echo "== chain $c"; python3 tools/ff_timeline.py $O/tl$c | tee $O/timeline_chain$c.txt
done
find $O -name "*.csv" -size +1M -delete
