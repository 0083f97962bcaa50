#!/bin/bash
# symmetric Gram partials (3 quadrants; sym) vs 4 quadrants (ld4 = previous commit)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02ad; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullframe.py tests/test_gpu_state_reuse.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in ld4 sym ld4 sym; do for f in 1 8; do
  WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so python bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames $f > $O/$v.$f.json 2> $O/$v.$f.err || { tail $O/$v.$f.err; exit 1; }
  python -c "import json; j=json.load(open('$O/$v.$f.json')); print('$v planes $f:', round(j['value'],1), 'fps, embed ms/plane', round(j['embed_ms_per_plane'],2))"
done; done
