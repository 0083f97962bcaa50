#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02ag; mkdir -p $O
WMHIP_LIB=$PWD/tools/bin/libwmhip_symdiag.so python - > $O/diag.log 2>&1 <<'PY'
import importlib, numpy as np
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
c = api.Context(0)
x = np.random.default_rng(0).integers(0, 256, (1080, 1920), dtype=np.uint8)
c.ref_sigma(x)
PY
sort $O/diag.log | uniq -c | sort -rn | head -12
