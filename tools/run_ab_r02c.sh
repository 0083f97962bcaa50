#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c; mkdir -p $O
B=tools/bin
python tools/ab_embed.py $B/libwmhip_base.so $B/libwmhip_nochk3.so $B/libwmhip_skip12.so $B/libwmhip_both.so $B/libwmhip_both11.so $B/libwmhip_s4_12.so $B/libwmhip_s4_11.so --rounds 9 > $O/ab.log 2>&1 || { tail -20 $O/ab.log; exit 1; }
cat $O/ab.log
