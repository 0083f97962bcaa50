#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in notriv triv; do for c in noise natural screen flat; do
  echo "== $v $c"; WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so timeout -k 10 200 python tools/quick_bench.py --content $c --frames 8 2>&1 | grep -E "^embed" | cut -c1-100
done; done
timeout -k 10 300 python tools/ab_embed.py tools/bin/libwmhip_notriv.so tools/bin/libwmhip_triv.so --rounds 9 2>&1 | cut -c1-150
