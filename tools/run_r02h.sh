#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02h; mkdir -p $O
for v in in256 in512 in1024; do
  for f in 1 8; do
    WMHIP_LIB=$PWD/tools/bin/libwmhip_$v.so python bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames $f > $O/$v.$f.json 2> $O/$v.$f.err || { tail $O/$v.$f.err; exit 1; }
    python -c "import json; j=json.load(open('$O/$v.$f.json')); print('$v', $f, 'planes:', round(j['value'],1), 'fps, embed ms/plane', round(j['embed_ms_per_plane'],2), 'parity', j.get('parity'))"
  done
done
