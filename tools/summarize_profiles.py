"""Turn the raw rocprofv3 CSVs that tools/collect_profiles.sh left under
gpurun_out/<tag>/ into the small judged files under profiles/.
    python tools/summarize_profiles.py r01_c
HBM bytes follow MI355X_MICROARCH.md (HBM section): separate --pmc passes,
FETCH_SIZE/WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE reports half the bytes
of wide coalesced reads (x2 correction; our 8 B/lane row loads are not separately
calibrated - the corrected figure is an upper bound)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_embed_tiles", "k_embed_fallback", "k_embed_one_small", "k_extract_tiles", "k_svd_tiles", "k_sigma_tiles", "k_detect_tiles")


def short(name):
    for k in KERNELS:
        if k + "<" in name or k + "(" in name:
            return k
    return None


def one(pattern):
    f = glob.glob(pattern)
    if not f:
        raise SystemExit("missing " + pattern)
    return f[0]


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", tag)
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    shutil.copy(one(base + "/stats/*/*_kernel_stats.csv"), os.path.join(prof, tag + "_kernel_stats.csv"))
    res = collections.defaultdict(dict)
    for r in csv.DictReader(open(one(base + "/stats/*/*_kernel_stats.csv"))):
        s = short(r["Name"])
        if s:
            res[s]["calls"] = int(r["Calls"]); res[s]["avg_us_kernel_trace"] = float(r["AverageNs"]) / 1e3
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("sq", "fetch", "write"):
        for r in csv.DictReader(open(one(base + f"/{sub}/*/*_counter_collection.csv"))):
            s = short(r["Kernel_Name"])
            if s:
                acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[s]["_dur_" + sub].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for s, d in acc.items():
        for c, v in d.items():
            res[s][c if not c.startswith("_dur_") else "avg_us_under_pmc_" + c[5:]] = \
                sum(v) / len(v) / (1e3 if c.startswith("_dur_") else 1)
    bench = json.load(open(os.path.join(base, "bench_unprofiled.json")))
    cfg = bench["config"]
    F, H, W = cfg["frames_per_rank"], cfg["height"], cfg["width"]
    e = res["k_embed_tiles"]
    out = dict(tag=tag, command=open(os.path.join(ROOT, "tools", "collect_profiles.sh")).read().split('CMD="')[1].split('"')[0],
               shape=dict(frames_per_launch=F, H=H, W=W), kernels=res, bench_line=bench)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        fetch_b = e["FETCH_SIZE"] * 1024 * 2
        write_b = e["WRITE_SIZE"] * 1024
        t = e["avg_us_under_pmc_sq"] * 1e-6
        clk = e["GRBM_GUI_ACTIVE"] / 8 / t
        n_waves = F * (((H // 8) * (W // 8) + 63) // 64)
        out["embed"] = dict(
            algorithmic_bytes_per_launch=3.0 * H * W * F,
            # the bench line's roofline.frac from THIS file: algorithmic bytes / kernel-trace average duration / 8 TB/s (the line itself
            # divides by HIP-event time un-profiled, a few % shorter)
            roofline_frac_kernel_trace=3.0 * H * W * F / (e["avg_us_kernel_trace"] * 1e-6) / 8e12,
            roofline_frac_bench_line=bench["roofline"]["frac"],
            fetch_bytes_x2_corrected=fetch_b, write_bytes=write_b,
            hbm_bytes_per_launch=fetch_b + write_b,
            traffic_over_algorithmic=(fetch_b + write_b) / (3.0 * H * W * F),
            valu_insts_per_wave=e["SQ_INSTS_VALU"] / n_waves,
            valu_busy_fraction=min(1.0, e["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * t * clk)),       # capped: the clock is an estimate (GRBM_GUI_ACTIVE)
            valu_busy_ratio_raw_uncapped=e["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * t * clk),
            effective_clock_GHz=clk / 1e9,
            wave_instr_per_s=e["SQ_INSTS_VALU"] / (e["avg_us_kernel_trace"] * 1e-6))
        json.dump(dict(hbm_bytes_per_launch_at_bench_shape=fetch_b + write_b,
                       frames_per_launch=F, H=H, W=W, source=f"profiles/{tag}_summary.json",
                       measured=__import__("datetime").date.today().isoformat(),
                       valu_busy_fraction=out["embed"]["valu_busy_fraction"],
                       effective_clock_GHz=out["embed"]["effective_clock_GHz"],
                       valu_insts_per_wave=out["embed"]["valu_insts_per_wave"]),
                  open(os.path.join(prof, "pmc_embed_latest.json"), "w"))
    x = res.get("k_extract_tiles", {})
    if "FETCH_SIZE" in x and "WRITE_SIZE" in x:
        nt = (H // 8) * (W // 8)
        # reads: stego P + Sc P/2 per plane, the shared pixel-domain factors 8 P once per launch; writes: float32 plane 4 P
        alg_r = F * (H * W + nt * 32) + 2 * nt * 256
        alg_w = F * H * W * 4
        out["extract"] = dict(algorithmic_read_bytes=alg_r, algorithmic_write_bytes=alg_w,
                              fetch_bytes_raw=x["FETCH_SIZE"] * 1024, fetch_bytes_x2_corrected=x["FETCH_SIZE"] * 2048,
                              write_bytes=x["WRITE_SIZE"] * 1024,
                              fetch_x2_over_algorithmic_reads=x["FETCH_SIZE"] * 2048 / alg_r,
                              avg_us_kernel_trace=x.get("avg_us_kernel_trace"))
    json.dump(out, open(os.path.join(prof, tag + "_summary.json"), "w"), indent=1)
    print(json.dumps(out.get("embed", {}), indent=1))
    print({k: v.get("avg_us_kernel_trace") for k, v in res.items()})


if __name__ == "__main__":
    main()
