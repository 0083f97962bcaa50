"""Time every BASELINE.json config on one GPU (device-resident, HIP events) with an
oracle parity check on a crop of full tiles.  Output: one JSON object per config
(stdout + gpurun_out/configs.json).  bench.py remains the contract benchmark; this
is the per-config evidence table quoted in DESIGN.md."""
import importlib, json, os, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
api = importlib.import_module("digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd.hostapi")
from oracle import wm_oracle as o

ctx = api.Context(0)
REPS = 10


def dev(arr):
    p = ctx.malloc(arr.nbytes); ctx.h2d(p, arr); return p


def timed(fn):
    fn(); ctx.sync()
    ctx.event_record(0)
    for _ in range(REPS):
        fn()
    ctx.event_record(1)
    return ctx.event_elapsed_ms(0, 1) / REPS


def run(name, H, W, n_planes, alpha, per_plane_wm, ops, K=8):
    nt = (H // 8) * (W // 8)
    rng = np.random.default_rng(1234)
    hosts = rng.integers(0, 256, (n_planes, H, W), dtype=np.uint8)
    n_wm = n_planes if per_plane_wm else 1
    wys = np.random.default_rng(4321).integers(0, 256, (n_wm, H, W)).astype(np.float32)
    d_host, d_wys = dev(hosts), dev(wys)
    d_stego = ctx.malloc(hosts.nbytes)
    d_U = ctx.malloc(n_wm * nt * 256); d_V = ctx.malloc(n_wm * nt * 256); d_S = ctx.malloc(n_wm * nt * 32)
    d_sc = ctx.malloc(n_planes * nt * 32); d_out = ctx.malloc(n_planes * H * W * 4); d_scores = ctx.malloc(n_planes * 8)
    sw_ps = nt * 8 if per_plane_wm else 0
    uv_ps = nt if per_plane_wm else 0
    res = dict(config=name, H=H, W=W, planes=n_planes, alpha=alpha, K=K)
    res["svd_watermark_ms"] = timed(lambda: ctx.svd_tiles_f32_dev(d_wys, d_U, d_S, d_V, n_wm, H, W, W, H * W))
    embed = lambda: ctx.embed_tiles_u8_dev(d_host, d_S, d_stego, d_sc, None, n_planes, H, W, W, H * W, sw_ps, alpha, K)
    P = n_planes * H * W
    if "embed" in ops:
        ms = timed(embed); res["embed_ms"] = ms; res["embed_GBps_algorithmic"] = 3.0 * P / ms / 1e6
    embed(); ctx.sync()
    if "extract" in ops:
        ms = timed(lambda: ctx.extract_tiles_u8_dev(d_stego, d_sc, d_U, d_V, d_out, n_planes, H, W, W, H * W, uv_ps, alpha, K))
        res["extract_ms"] = ms; res["extract_GBps_algorithmic"] = 13.5 * P / ms / 1e6
    if "detect" in ops:
        ms = timed(lambda: ctx.detect_tiles_u8_dev(d_stego, d_sc, d_S, d_scores, n_planes, H, W, W, H * W, sw_ps, alpha))
        res["detect_ms"] = ms; res["detect_GBps_algorithmic"] = 2.0 * P / ms / 1e6
    tot = sum(res.get(k, 0.0) for k in ("embed_ms", "extract_ms", "detect_ms"))
    res["frames_per_s"] = n_planes / tot * 1e3
    # parity on a crop of full tiles of plane 0 (tiles are independent)
    stego = np.empty_like(hosts); ctx.d2h(stego, d_stego)
    ch, cw = min(H, 64), min(W, 128)
    ref = o.embed_plane(hosts[0, :ch, :cw].astype(np.float32), wys[0, :ch, :cw], alpha, 0.0, 8, k_floor=K)
    d = np.abs(stego[0, :ch, :cw].astype(int) - ref["stego"].astype(int))
    res["parity_crop_max_lsb"] = int(d.max()); res["psnr_gpu_plane0"] = o.psnr(hosts[0], stego[0])
    if "detect" in ops:
        sc = np.zeros(n_planes, np.float64); ctx.d2h(sc, d_scores); res["detect_score_plane0"] = float(sc[0])
    ctx.check_status()
    for p in (d_host, d_wys, d_stego, d_U, d_V, d_S, d_sc, d_out, d_scores):
        ctx.free(p)
    print(json.dumps(res), flush=True)
    return res


out = []
out.append(run("cfg1 512x512 gray host (GPU counterpart of the CPU plumbing case)", 512, 512, 1, 0.12, False, ("embed", "extract", "detect")))
out.append(run("cfg2 1920x1080 Y embed+extract", 1080, 1920, 1, 0.15, False, ("embed", "extract")))
out.append(run("cfg3 3840x2160 B,G,R colour-watermark embed", 2160, 3840, 3, 0.18, True, ("embed",)))
out.append(run("cfg4 256 frames 1080p (one rank's view: all 256; 32 per rank at 8 GPUs)", 1080, 1920, 256, 0.15, False, ("embed",)))
out.append(run("cfg4 32 frames 1080p (one rank's share at 8 GPUs)", 1080, 1920, 32, 0.15, False, ("embed",)))
for K in (1, 2, 4, 6, 8):
    out.append(run(f"cfg5 7680x4320 embed+extract+detect, mid-band K={K} (k_floor sweep)", 4320, 7680, 1, 0.15, False,
                   ("embed", "extract", "detect"), K=K))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/configs.json", "w"), indent=1)
