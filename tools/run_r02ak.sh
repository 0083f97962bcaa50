#!/bin/bash
# cross step with carried LDS offsets, compile-time buffers, bpermute coefficients (xstep) vs previous (sym)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02ak; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fullframe.py tests/test_gpu_state_reuse.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in sym xrl sym xrl; do for f in 1 8; do
  WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so python bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames $f > $O/$v.$f.json 2> $O/$v.$f.err || { tail $O/$v.$f.err; exit 1; }
  python -c "import json; j=json.load(open('$O/$v.$f.json')); print('$v planes $f:', round(j['value'],1), 'fps, embed ms/plane', round(j['embed_ms_per_plane'],2))"
done; done
WMHIP_LIB=$PWD/tools/bin/libwmhip_xrldiag.so python - > $O/diag.log 2>&1 <<'PY'
import importlib, numpy as np
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
c = api.Context(0)
x = np.random.default_rng(0).integers(0, 256, (1080, 1920), dtype=np.uint8)
c.ref_sigma(x)
PY
sort $O/diag.log | uniq -c | sort -rn | head -6
