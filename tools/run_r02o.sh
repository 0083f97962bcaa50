#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02o; mkdir -p $O
export WM_RF_QUEUES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ffstats -- python3 bench.py --mode fullframe --steps 2 --cpu-frames 0 > $O/ffstats.json 2> $O/ffstats.err || { tail $O/ffstats.err; exit 1; }
head -8 $O/ffstats/*/*_kernel_stats.csv | cut -c1-60,150-260
