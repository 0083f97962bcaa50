"""gpurun_out/<tag>/ (tools/collect_profiles_pixel.sh) -> profiles/<tag>_pixel_kernel_stats.csv + profiles/<tag>_pixel_summary.json:
per kernel the rocprofv3 average duration, the algorithmic bytes of one launch (tools/pixel_roofline.py's figures), bytes / time /
8 TB/s, and the HBM traffic from the PMC passes (FETCH_SIZE x 2 on gfx950 for wide coalesced reads - MI355X_MICROARCH.md, HBM
section; narrower accesses are uncalibrated, so the corrected fetch figure is an upper bound - and WRITE_SIZE, both in KB).
    python tools/summarize_profiles_pixel.py r03_p"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
base = os.path.join(ROOT, "gpurun_out", tag)
ev = json.load(open(os.path.join(base, "events.json")))
F, H, W = ev["frames"], ev["H"], ev["W"]
n = H * W


def one(p):
    f = glob.glob(p)
    if not f:
        raise SystemExit("missing " + p)
    return f[0]


# kernel-name fragment -> (label, algorithmic bytes of ONE launch, launches that share the figure)
K = {
    "k_permute<unsigned char>": ("k_permute<u8> (gather, single:66-72)", 9.0 * n * F),
    "k_permute<float>": ("k_permute<f32> (gather)", 12.0 * n * F),
    "k_unpermute(": ("k_unpermute (scatter, single:74-80)", 12.0 * n * F),
    "k_minmax(": ("k_minmax (single:221)", 4.0 * n),
    "k_normalize_u8": ("k_normalize_u8 (single:221-222)", 5.0 * n),
    "k_minmax_planes": ("k_minmax_planes (routed chain)", 4.0 * n * F),
    "k_route_p1": ("k_route_p1 (routed unscramble, pass 1)", 9.0 * n * F),
    "k_route_p2": ("k_route_p2 (routed unscramble, pass 2)", 4.0 * n * F),
    "k_route_ga": ("k_route_ga (routed scramble, pass A)", 4.0 * n * F),
    "k_route_gb": ("k_route_gb (routed scramble, pass B)", 9.0 * n * F),
    "k_color<0>": ("k_color BGR->YCrCb (single:21-24)", 6.0 * n * F),
    "k_color<3>": ("k_color BGR->Y", 4.0 * n * F),
    "k_color<4>": ("k_color replace Y -> BGR (single:26-30)", 7.0 * n * F),
    "k_sqdiff_u8": ("k_sqdiff_u8 (PSNR, single:38-42)", 6.0 * n * F),
    "k_ssim": ("k_ssim (single:44-57)", None),
}


def key(name):
    name = name.replace("(anonymous namespace)::", "")
    for k in K:
        if k in name:
            return k
    if "k_color<(" in name:                       # enum spelled out by the demangler
        for code, frag in (("0", "k_color<0>"), ("3", "k_color<3>"), ("4", "k_color<4>")):
            if f")" + code + ">" in name:
                return frag
    return None


os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
shutil.copy(one(base + "/stats/*/*_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_pixel_kernel_stats.csv"))
res = collections.OrderedDict()
for r in csv.DictReader(open(one(base + "/stats/*/*_kernel_stats.csv"))):
    k = key(r["Name"])
    if k:
        d = res.setdefault(k, {"label": K[k][0], "calls": 0, "total_ns": 0.0})
        d["calls"] += int(r["Calls"]); d["total_ns"] += float(r["TotalDurationNs"])
for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(one(base + f"/{sub}/*/*_counter_collection.csv"))):
        k = key(r["Kernel_Name"])
        if k and r["Counter_Name"] == cname:
            acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if k in res:
            res[k][cname + "_KB_per_launch_raw"] = sum(v) / len(v)
out = []
for k, d in res.items():
    avg_us = d["total_ns"] / d["calls"] / 1e3
    alg = K[k][1]
    row = {"kernel": d["label"], "calls": d["calls"], "avg_us_kernel_trace": avg_us}
    if k == "k_ssim":
        alg = 2.0 * n                      # one launch = one plane pair; uint8/uint8 and uint8/float32 launches are mixed in the stats row
        row["note"] = "VALU-bound stencil (11x11 Gaussian of 5 fields); bytes are the uint8 pair's"
    row["algorithmic_bytes_per_launch"] = alg
    row["GBps"] = alg / avg_us / 1e3
    row["frac_of_8TBps"] = row["GBps"] / 8000.0
    f = d.get("FETCH_SIZE_KB_per_launch_raw"); w = d.get("WRITE_SIZE_KB_per_launch_raw")
    if f is not None and w is not None:
        row["hbm_traffic_bytes_per_launch"] = {"fetch_raw": f * 1024, "fetch_x2_gfx950": 2 * f * 1024, "write": w * 1024,
                                               "total_corrected": 2 * f * 1024 + w * 1024,
                                               "over_algorithmic": (2 * f * 1024 + w * 1024) / alg}
    out.append(row)
json.dump({"tag": tag, "shape": {"H": H, "W": W, "planes_per_launch": F}, "command": "python3 tools/pixel_roofline.py --frames 8 --reps 10",
           "hip_event_table": ev["rows"], "kernels": out}, open(os.path.join(ROOT, "profiles", tag + "_pixel_summary.json"), "w"), indent=1)
for r in out:
    t = r.get("hbm_traffic_bytes_per_launch")
    print(f"{r['kernel']:46s} {r['avg_us_kernel_trace']:9.1f} us  {r['GBps']:7.0f} GB/s = {100 * r['frac_of_8TBps']:5.1f} %"
          + (f"   traffic {t['total_corrected'] / 1e6:8.1f} MB = {t['over_algorithmic']:.2f} x algorithmic" if t else ""))
