#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02d; mkdir -p $O
B=tools/bin
python tools/ab_embed.py $B/libwmhip_base.so $B/libwmhip_s4_12.so $B/libwmhip_r4_12.so $B/libwmhip_r4_11.so $B/libwmhip_r4_11s6.so $B/libwmhip_r4_11s5.so --rounds 9 > $O/ab.log 2>&1 || { tail -20 $O/ab.log; exit 1; }
cat $O/ab.log
