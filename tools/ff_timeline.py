#!/usr/bin/env python3
"""GPU timeline of a full-frame batch from a rocprofv3 --kernel-trace CSV: per kernel average duration, and for the
block-Jacobi phase (first k_rf_gram .. last k_rf_apply of the LAST call) wall time, union of busy intervals, sum of
kernel durations: idle = wall - union (launch / dependency gaps), overlap = sum / union (queues running side by side).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/prof_ff_batch.py 8
    python3 tools/ff_timeline.py gpurun_out/tl
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel_trace.csv under " + d
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"\bk_\w+", r["Kernel_Name"]) or re.search(r"\w+", r["Kernel_Name"])).group(0), r.get("Queue_Id", "?")))
    rows.sort()
    jac = [i for i, r in enumerate(rows) if "k_rf_gram" in r[2] or "k_rf_apply" in r[2] or "k_rf_inner" in r[2]]
    # split into calls: a gap > 2 ms between consecutive Jacobi kernels starts a new call
    calls, cur = [], [jac[0]]
    for a, b in zip(jac, jac[1:]):
        if rows[b][0] - rows[a][1] > 2_000_000:
            calls.append(cur); cur = []
        cur.append(b)
    calls.append(cur)
    sel = [rows[i] for i in max(calls, key=len)]
    t0, t1 = sel[0][0], max(r[1] for r in sel)
    busy, end = 0, t0
    for s, e, _, _ in sel:
        if e > end:
            busy += e - max(s, end); end = e
    tot = sum(e - s for s, e, _, _ in sel)
    per = defaultdict(list)
    for s, e, n, q in sel:
        per[n].append(e - s)
    queues = sorted({r[3] for r in sel})
    print(f"jacobi phase: {len(sel)} kernels on queues {queues}: wall {(t1 - t0) / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f} ms, "
          f"sum of kernels {tot / 1e6:.2f} ms -> idle {(t1 - t0 - busy) / 1e6:.2f} ms, overlap factor {tot / busy:.2f}")
    for n, v in sorted(per.items()):
        print(f"  {n:24s} n {len(v):5d}  avg {sum(v) / len(v) / 1e3:7.1f} us  total {sum(v) / 1e6:7.2f} ms")
    # gaps between consecutive kernels of one queue
    for q in queues:
        ks = [r for r in sel if r[3] == q]
        gaps = [b[0] - a[1] for a, b in zip(ks, ks[1:])]
        gaps.sort()
        print(f"  queue {q}: {len(ks)} kernels, median gap {gaps[len(gaps) // 2] / 1e3:.1f} us, mean {sum(gaps) / len(gaps) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
