#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in fb2 fb1; do for c in noise screen flat; do
  echo "== $v $c"; WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so timeout -k 10 200 python tools/quick_bench.py --content $c --frames 8 2>&1 | grep -E "^embed|^extract_px|^detect" | cut -c1-100
done; done
