#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02m; mkdir -p $O
B=tools/bin
timeout -k 10 300 python tools/sigma_skip_check.py $B/libwmhip_sk0.so $B/libwmhip_sk6.so $B/libwmhip_sk7.so $B/libwmhip_sk75.so $B/libwmhip_sk8.so > $O/check.log 2>&1 || { tail -20 $O/check.log; exit 1; }
cat $O/check.log
timeout -k 10 300 python tools/ab_embed.py $B/libwmhip_sk0.so $B/libwmhip_sk6.so $B/libwmhip_sk7.so $B/libwmhip_sk75.so $B/libwmhip_sk8.so --rounds 7 > $O/ab.log 2>&1 || { tail -20 $O/ab.log; exit 1; }
cut -c1-110 $O/ab.log
