#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02f; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 bench.py --mode fullframe --steps 1 --cpu-frames 0 --ff-frames 1 > $O/sq.json 2> $O/sq.err || { tail $O/sq.err; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 bench.py --mode fullframe --steps 1 --cpu-frames 0 --ff-frames 1 > $O/sq2.json 2> $O/sq2.err || { tail $O/sq2.err; exit 1; }
python - <<'PY'
import csv, glob, collections
for sub in ("sq", "sq2"):
    f = glob.glob(f"gpurun_out/r02f/{sub}/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "")[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in ("k_rf_inner", "k_rf_apply", "k_rf_gram"):
        for kk, d in acc.items():
            if k in kk:
                print(sub, kk, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(d["dur_ns"]))
PY
