timeout -k 10 900 python -m pytest tests/test_gpu_fullframe.py tests/test_gpu_dropin.py tests/test_capi.py -x -q 2>&1 | tail -3
python tools/array_latency.py 2>&1 | grep "tile=None"
