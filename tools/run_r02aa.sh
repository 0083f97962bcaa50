#!/bin/bash
# where an 8-plane full-frame embed's time goes on the GPU timeline (1 and 2 queues)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02aa; mkdir -p $O
for q in 1 2 3; do
  export WM_RF_QUEUES=$q
  rocprofv3 --kernel-trace --output-format csv -d $O/tl$q -- python3 tools/prof_ff_batch.py 8 > $O/tl$q.log 2>&1 || { tail $O/tl$q.log; exit 1; }
  echo "== queues $q"; python3 tools/ff_timeline.py $O/tl$q | tee $O/timeline$q.txt
done
export WM_RF_QUEUES=1
rocprofv3 --kernel-trace --output-format csv -d $O/tl1p -- python3 tools/prof_ff_batch.py 1 > $O/tl1p.log 2>&1 || { tail $O/tl1p.log; exit 1; }
echo "== 1 plane"; python3 tools/ff_timeline.py $O/tl1p | tee $O/timeline1p.txt
find $O -name "*.csv" -size +1M -delete
