#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02x; mkdir -p $O
for c in 2e-4 5e-4 1e-3 2e-3; do
  WM_RF_CONV_SIGMA=$c python bench.py --mode fullframe --steps 2 --ff-frames 8 > $O/c$c.json 2> $O/c$c.err || { tail $O/c$c.err; exit 1; }
  python -c "import json; j=json.load(open('$O/c$c.json')); print('conv $c:', round(j['value'],1), 'fps', j['roofline']['note'][:9], j['parity'])"
done
