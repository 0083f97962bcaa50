#!/bin/bash
# gpurun -- 'bash tools/run_ab_r02a.sh'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02a; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
B=tools/bin
python tools/ab_embed.py $B/libwmhip_base.so $B/libwmhip_oldmap.so $B/libwmhip_skip12.so $B/libwmhip_skip13.so $B/libwmhip_nochk3.so $B/libwmhip_w4.so $B/libwmhip_both.so > $O/ab.log 2>&1 || { tail -20 $O/ab.log; exit 1; }
cat $O/ab.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/ab_embed.py $B/libwmhip_base.so $B/libwmhip_oldmap.so --rounds 1 --reps 2 > $O/fetch.log 2>&1 || { tail -20 $O/fetch.log; exit 1; }
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r02a/fetch/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_extract_tiles" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[(r["Process_Id"] if "Process_Id" in r else "", r["Kernel_Name"][:60], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
rows = list(csv.DictReader(open(f)))
ex = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in rows if "k_extract_tiles" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
ex.sort()
print("extract FETCH_SIZE (KB) per dispatch in order:", [round(v) for _, v in ex])
PY
