#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02e; mkdir -p $O
python -m pytest tests/test_gpu_fullframe.py -m gpu -x -q -k "device_pointer" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cat $O/bench.json
python bench.py --mode fullframe --steps 3 --cpu-frames 0 > $O/bench_ff.json 2> $O/bench_ff.err || { tail $O/bench_ff.err; exit 1; }
cat $O/bench_ff.json
python bench.py --mode fullframe --steps 3 --cpu-frames 0 --ff-frames 1 > $O/bench_ff1.json 2> $O/bench_ff1.err || { tail $O/bench_ff1.err; exit 1; }
cat $O/bench_ff1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ffstats -- python3 bench.py --mode fullframe --steps 2 --cpu-frames 0 > $O/ffstats.json 2> $O/ffstats.err || { tail $O/ffstats.err; exit 1; }
head -12 $O/ffstats/*/*_kernel_stats.csv | cut -c1-200
