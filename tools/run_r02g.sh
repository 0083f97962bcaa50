#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02g; mkdir -p $O
python -m pytest tests/test_gpu_fullframe.py tests/test_gpu_state_reuse.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python bench.py --mode fullframe --steps 3 --cpu-frames 0 > $O/bench_ff.json 2> $O/bench_ff.err || { tail $O/bench_ff.err; exit 1; }
python -c "import json; j=json.load(open('$O/bench_ff.json')); print('8 planes:', round(j['value'],1), 'fps, embed ms/plane', round(j['embed_ms_per_plane'],2), 'frac', round(j['roofline']['frac'],3))"
python bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames 1 > $O/bench_ff1.json 2> $O/bench_ff1.err || { tail $O/bench_ff1.err; exit 1; }
python -c "import json; j=json.load(open('$O/bench_ff1.json')); print('1 plane:', round(j['value'],1), 'fps, embed ms/plane', round(j['embed_ms_per_plane'],2))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ffstats -- python3 bench.py --mode fullframe --steps 2 --cpu-frames 0 --ff-frames 1 > $O/ffstats.json 2> $O/ffstats.err || { tail $O/ffstats.err; exit 1; }
head -5 $O/ffstats/*/*_kernel_stats.csv | cut -c1-150
