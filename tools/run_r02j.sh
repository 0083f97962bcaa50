#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02j; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
B=tools/bin
timeout -k 10 300 python tools/ab_embed.py $B/libwmhip_noasm.so $B/libwmhip_asm3.so $B/libwmhip_asm3b.so $B/libwmhip_asm4b.so --rounds 9 > $O/ab.log 2>&1 || { tail -20 $O/ab.log; exit 1; }
cat $O/ab.log
