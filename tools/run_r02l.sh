#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02l; mkdir -p $O
B=tools/bin
timeout -k 10 300 python tools/ab_embed.py $B/libwmhip_sk0.so $B/libwmhip_sk6.so $B/libwmhip_sk6f3.so $B/libwmhip_sk5.so --rounds 9 > $O/ab.log 2>&1 || { tail -20 $O/ab.log; exit 1; }
cat $O/ab.log
