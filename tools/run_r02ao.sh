#!/bin/bash
# literal chain without scratch memory (noscr: packed row-pass DCT, pattern added in the packed layout) vs before (cst)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02ao
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02ao/pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02ao/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in cst noscr; do for c in noise natural screen flat letterbox; do
  echo "== $v $c"; WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so timeout -k 10 200 python tools/quick_bench.py --content $c --frames 8 2>&1 | grep -E "^(embed |content)" | cut -c1-110
done; done 2>&1 | tee gpurun_out/r02ao/content.log
timeout -k 10 600 python tools/ab_embed.py tools/bin/libwmhip_cst.so tools/bin/libwmhip_noscr.so --rounds 9 2>&1 | cut -c1-130 | tail -3
