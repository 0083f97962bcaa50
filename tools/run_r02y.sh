#!/bin/bash
# with-V Jacobi as a generated gfx950 stream (asmv) against the C++ form (cppv): GPU tests, then timing by content
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02y
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02y/pytest.log 2>&1; rc=$?; tail -5 gpurun_out/r02y/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in cppv asmv; do for c in noise natural screen flat; do
  echo "== $v $c"; WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so timeout -k 10 200 python tools/quick_bench.py --content $c --frames 8 2>&1 | grep -E "^(embed|svd)" | cut -c1-110
done; done 2>&1 | tee gpurun_out/r02y/content.log
