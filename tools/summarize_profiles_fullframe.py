"""Condense the rocprofv3 CSVs of tools/collect_profiles_fullframe.sh (run on the GPU box, where the raw
counter files are too large to bring back) into one small JSON per batch size:
    python3 tools/summarize_profiles_fullframe.py gpurun_out/<tag> <F>  ->  gpurun_out/<tag>/summary<F>.json
Per kernel: calls, average / total duration from --kernel-trace --stats, and per-launch averages of every PMC
counter collected (FETCH_SIZE and WRITE_SIZE are KiB on this stack; MI355X_MICROARCH.md: FETCH_SIZE reports
half the bytes of wide coalesced reads on gfx950 - the x2 figure is given beside the raw one)."""
import collections
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    if not m:
        return "__amd_copy" if "rocclr" in name else None
    return m.group(1) + (m.group(2) or "")


def main():
    base, F = sys.argv[1], sys.argv[2]
    res = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(base, f"stats{F}", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r["Name"])
            if s:
                d = res[s]
                d["calls"] = d.get("calls", 0) + int(r["Calls"])
                d["total_ms"] = d.get("total_ms", 0.0) + float(r["TotalDurationNs"]) / 1e6
    for s, d in res.items():
        d["avg_us"] = d["total_ms"] * 1e3 / max(1, d["calls"])
    for sub in ("fetch", "write", "sq", "lds"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        dur = collections.defaultdict(float)
        for f in glob.glob(os.path.join(base, f"{sub}{F}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                s = short(r["Kernel_Name"])
                if not s:
                    continue
                acc[s][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in cnt[s]:
                    cnt[s].add(r["Dispatch_Id"])
                    dur[s] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for s, d in acc.items():
            n = max(1, len(cnt[s]))
            for c, v in d.items():
                res[s][c + "_per_launch"] = v / n
            res[s][f"avg_us_under_pmc_{sub}"] = dur[s] / n / 1e3
            res[s][f"launches_{sub}"] = n
    for s, d in res.items():
        if "FETCH_SIZE_per_launch" in d:
            d["fetch_bytes_raw"] = d["FETCH_SIZE_per_launch"] * 1024
            d["fetch_bytes_x2"] = d["FETCH_SIZE_per_launch"] * 2048
        if "WRITE_SIZE_per_launch" in d:
            d["write_bytes"] = d["WRITE_SIZE_per_launch"] * 1024
    bench = None
    try:
        bench = json.loads(open(os.path.join(base, f"bench{F}.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    total = sum(d.get("total_ms", 0.0) for d in res.values())
    out = {"planes_per_step": int(F), "kernels": dict(sorted(res.items(), key=lambda kv: -kv[1].get("total_ms", 0.0))),
           "sum_kernel_ms": total, "bench_line_unprofiled": bench}
    json.dump(out, open(os.path.join(base, f"summary{F}.json"), "w"), indent=1)
    for s, d in out["kernels"].items():
        if d.get("total_ms", 0) > 0.01 * total:
            print(f"{s:34s} calls {d.get('calls', 0):6d} avg {d.get('avg_us', 0):8.1f} us  {100 * d.get('total_ms', 0) / total:5.1f} %"
                  f"  fetch(x2) {d.get('fetch_bytes_x2', float('nan')) / 1e6:8.2f} MB  write {d.get('write_bytes', float('nan')) / 1e6:8.2f} MB")


if __name__ == "__main__":
    main()
