#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02i; mkdir -p $O
for v in diag512 diag1024; do
WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so python - > $O/$v.log 2>&1 <<'PY'
import importlib, numpy as np
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
c = api.Context(0)
x = np.random.default_rng(0).integers(0, 256, (1080, 1920), dtype=np.uint8)
c.ref_sigma(x)
PY
echo $v; sort $O/$v.log | uniq -c | sort -rn | head -3; sed -n '1,3p;100,102p;400,402p' $O/$v.log
done
