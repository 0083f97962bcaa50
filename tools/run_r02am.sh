#!/bin/bash
# constant tiles in closed form (cst) vs the literal chain for every rank-deficient tile (base): GPU tests, then timing by content
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02am
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02am/pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02am/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in base cst; do for c in noise natural screen flat letterbox; do
  echo "== $v $c"; WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so timeout -k 10 200 python tools/quick_bench.py --content $c --frames 8 2>&1 | grep -E "^(embed |content)" | cut -c1-110
done; done 2>&1 | tee gpurun_out/r02am/content.log
