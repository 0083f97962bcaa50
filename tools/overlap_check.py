"""From a rocprofv3 kernel_trace.csv: how much of each k_rf_inner's span is overlapped by
kernels of the other queue (two-queue batched Jacobi)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_rf_" in r["Kernel_Name"]]
print("columns:", list(rows[0].keys()))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:12]) for r in rows]
ev.sort()
qs = sorted(set(e[2] for e in ev))
print("queues:", qs, "kernels:", len(ev))
mid = ev[len(ev) // 2: len(ev) // 2 + 24]
t0 = mid[0][0]
for s, e, q, n in mid:
    print(f"q{qs.index(q)} {n:12s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us")
span = (ev[-1][1] - ev[0][0]) / 1e3
busy = sum(e - s for s, e, _, _ in ev) / 1e3
print(f"span {span:.0f} us, sum of kernel durations {busy:.0f} us, ratio {busy / span:.2f}")
