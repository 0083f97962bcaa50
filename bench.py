"""bench.py - frames/sec for tile-mode embed + extract on 4K Y-plane frames.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ...`
   or started bare - then this process, before it touches the GPU, starts the N
   ranks itself as child processes and relays rank 0's JSON line; one rank per
   GPU over RCCL either way)

One *step* = one pass of the hot path over one batch of synthetic frames that
are already resident in HBM: per rank, F frames of 2160x3840 uint8 Y ->
K1 embed (stego + Sc) -> K2+K4 extract (scrambled-watermark estimate), after
the watermark's singular values were broadcast from rank 0 (RCCL; the one
exchange step of the path).  Weak scaling: every rank processes F frames.
Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: ``roofline`` (embed kernel, algorithmic bytes / HIP-event time) and
``cpu_baseline`` (the NumPy oracle timed on this box's host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd"

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
VALU_PEAK_LANE_OPS = 73e12     # measured v_pk_fma_f32 lane-op/s ceiling on this part (tools/ubench_valu)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64,
                    help="frames per rank per step (64 since round 3b: a launch of 32 frames is 21 rounds of resident waves and "
                         "loses 5 %% to its last one - 19.9 k frames/s at 32, 20.9 k at 64, 20.7 k at 128 / 256, profiles/r03z_frames_sweep.log)")
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--alpha", type=float, default=0.15)
    ap.add_argument("--cpu-frames", type=int, default=3, help="frames in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--mode", choices=("tile", "fullframe"), default="tile",
                    help="tile: the 8x8 hot path (contract default); fullframe: the reference's own "
                         "semantics (one dense SVD per plane), secondary workload")
    ap.add_argument("--ff-frames", type=int, default=64,
                    help="full-frame section: planes per rank per step (round 3: 16).  The two-level block Jacobi with its split-f16 "
                         "products keeps gaining up to ~64 planes per call: 180 / 201 / 214 / 232 / 242 frames/s at 16 / 24 / 32 / 48 / 64 "
                         "(the flat tournament: 138-142 throughout); the 16-plane figure is still reported as value_at_16_planes")
    ap.add_argument("--ff-height", type=int, default=1080)
    ap.add_argument("--ff-width", type=int, default=1920)
    ap.add_argument("--no-fullframe", action="store_true", help="skip the full-frame section of the default line")
    ap.add_argument("--quick", action="store_true",
                    help="the contract line only: skip the end-to-end, pool / reference-semantics CPU and full-frame sections")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the rocprofv3 --pmc child passes that measure roofline.traffic / roofline.valu in this run "
                         "(the committed profile is replayed instead, and labelled so)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: initialise the process group (RCCL, world size 1) and issue the per-step broadcast of the "
                         "watermark's singular values all the same - `bcast_ms_per_step` in the line - so that the first multi-GPU "
                         "run is not also RCCL's first run")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses GPU 0 (with --backend gloo on a 1-GPU box)")
    return ap.parse_args()


def cpu_baseline(frames_u8, wys, alpha, stego_gpu, sc_gpu, wm_gpu):
    """Oracle (NumPy restatement, tile-mode) embed+extract on a bounded sample
    of the same workload, timed on this box's host cores."""
    from oracle import wm_oracle as o

    n = frames_u8.shape[0]
    H, W = frames_u8.shape[1:]
    t0 = time.perf_counter(); c0 = time.process_time()
    wm_svd = o.watermark_decompose(wys, 8)           # once per watermark, like the GPU path
    t_wm = time.perf_counter() - t0
    worst_lsb, worst_sig, worst_wm = 0, 0.0, 0.0
    psnr_cpu, psnr_gpu = [], []
    t1 = time.perf_counter()
    for i in range(n):
        e = o.embed_plane(frames_u8[i].astype(np.float32), wys, alpha, 0.6, tile=8, wm_svd=wm_svd)
        wm_cpu = o.extract_plane(e["stego"].astype(np.float32), e["Sc"], e["Uw"], e["Vwt"], alpha, 0.6, H, W, 8)
        worst_wm = max(worst_wm, float(np.abs(wm_cpu - wm_gpu[i]).max() / max(float(np.abs(wm_cpu).max()), 1e-30)))
        d = np.abs(e["stego"].astype(np.int16) - stego_gpu[i].astype(np.int16))
        worst_lsb = max(worst_lsb, int(d.max()))
        rel = np.max(np.abs(sc_gpu[i] - e["Sc"]) / np.maximum(e["Sc"][..., :1], 1e-30))
        worst_sig = max(worst_sig, float(rel))
        psnr_cpu.append(o.psnr(frames_u8[i], e["stego"]))
        psnr_gpu.append(o.psnr(frames_u8[i], stego_gpu[i]))
    wall = time.perf_counter() - t1
    cores = max(1, round((time.process_time() - c0) / max(time.perf_counter() - t0, 1e-9)))
    return dict(value=n / wall, unit="frames/s", cores=int(cores), kind="port",
                sample=f"{n} frames {W}x{H} Y, NumPy oracle tile-mode embed+extract "
                       f"(watermark SVD {t_wm:.2f}s once, excluded like on the GPU)",
                host_cpus=os.cpu_count()), \
        dict(stego_max_lsb=worst_lsb, sigma_max_rel=worst_sig, extract_max_rel_to_range=worst_wm,
             psnr_cpu=float(np.mean(psnr_cpu)), psnr_gpu=float(np.mean(psnr_gpu)))


_POOL = {}


def _pool_init(H, W):
    """worker start-up (untimed): the watermark's tile SVD, once per worker like once per video on the GPU"""
    from oracle import wm_oracle as o
    _POOL["o"] = o
    _POOL["wys"] = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    _POOL["svd"] = o.watermark_decompose(_POOL["wys"], 8)


def _pool_ready(_):
    return os.getpid()


def _pool_frame(args):
    frame, alpha = args
    o = _POOL["o"]
    H, W = frame.shape
    e = o.embed_plane(frame.astype(np.float32), _POOL["wys"], alpha, 0.6, tile=8, wm_svd=_POOL["svd"])
    o.extract_plane(e["stego"].astype(np.float32), e["Sc"], e["Uw"], e["Vwt"], alpha, 0.6, H, W, 8)
    return int(e["stego"][0, 0])


def cpu_baseline_pool(frames_u8, alpha):
    """The same tile-mode oracle under a process pool over frames: one frame per worker, as many workers as this
    process may run on (capped at 16 - the GPU box's CPU share per GPU).  Workers are spawned (the parent holds a
    HIP context), start-up and the per-worker watermark SVD are outside the timed region."""
    import multiprocessing as mp
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nw = max(1, min(avail, 16, frames_u8.shape[0]))
    H, W = frames_u8.shape[1:]
    with mp.get_context("spawn").Pool(nw, initializer=_pool_init, initargs=(H, W)) as pool:
        pool.map(_pool_ready, range(4 * nw), chunksize=1)                 # every worker is up and initialised
        t0 = time.perf_counter()
        pool.map(_pool_frame, [(frames_u8[i], alpha) for i in range(nw)], chunksize=1)
        wall = time.perf_counter() - t0
    return dict(value=nw / wall, unit="frames/s", cores=nw, kind="port",
                sample=f"{nw} frames, one per worker process: NumPy oracle tile-mode embed+extract", host_cpus=os.cpu_count(),
                cpus_available=avail)


def cpu_reference_semantics(frame_u8, wys, alpha):
    """What the REFERENCE itself does to one frame of this size (single:172-177, 205-218: one dense DCT + SVD of the
    whole plane, twice per embed - host and watermark - and once per extract), through the oracle's tile=None path;
    LAPACK threads as NumPy finds them."""
    from oracle import wm_oracle as o
    H, W = frame_u8.shape
    t0 = time.perf_counter(); c0 = time.process_time()
    e = o.embed_plane(frame_u8.astype(np.float32), wys, alpha, 0.6, None)
    t1 = time.perf_counter()
    o.extract_plane(e["stego"].astype(np.float32), e["Sc"], e["Uw"], e["Vwt"], alpha, 0.6, H, W, None)
    wall = time.perf_counter() - t0
    cores = max(1, round((time.process_time() - c0) / wall))
    return dict(value=1.0 / wall, unit="frames/s", cores=int(cores), kind="port", embed_s=t1 - t0, extract_s=wall - (t1 - t0),
                sample=f"1 frame {W}x{H}: the reference's own full-plane path (oracle tile=None) - embed incl. the watermark "
                       f"SVD + extract", host_cpus=os.cpu_count())


def pcie_roof(torch, dev, h2d_bytes, d2h_bytes, reps=6, attempts=3):
    """Plain pinned-memory copies of one batch's bytes in each direction, alone and both at once on two streams:
    what the link of THIS box gives, so that a slow box is distinguishable from a slow pipeline.  Best of `attempts`
    stream pairs (see end_to_end_section on why the pair matters)."""
    h_up = torch.empty(h2d_bytes, dtype=torch.uint8).pin_memory(); d_up = torch.empty(h2d_bytes, dtype=torch.uint8, device=dev)
    h_dn = torch.empty(d2h_bytes, dtype=torch.uint8).pin_memory(); d_dn = torch.empty(d2h_bytes, dtype=torch.uint8, device=dev)

    def run(s_up, s_dn, up, dn):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            if up:
                with torch.cuda.stream(s_up):
                    d_up.copy_(h_up, non_blocking=True)
            if dn:
                with torch.cuda.stream(s_dn):
                    h_dn.copy_(d_dn, non_blocking=True)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    t_up = t_dn = t_both = float("inf")
    for _ in range(attempts):
        s_up, s_dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        run(s_up, s_dn, True, True)
        t_up = min(t_up, run(s_up, s_dn, True, False)); t_dn = min(t_dn, run(s_up, s_dn, False, True))
        t_both = min(t_both, run(s_up, s_dn, True, True))
    return {"h2d": h2d_bytes / t_up / 1e9, "d2h": d2h_bytes / t_dn / 1e9,
            "h2d_concurrent": h2d_bytes / t_both / 1e9, "d2h_concurrent": d2h_bytes / t_both / 1e9,
            "bytes": {"h2d": h2d_bytes, "d2h": d2h_bytes}}


def end_to_end_section(torch, api, dev, H, W, alpha, Sw, Ux, Vxt, idx, F=8, batches=15, nbuf=3, max_stream_sets=4):
    """PCIe-inclusive tile-mode embed + FULL extract: pinned host frames -> H2D -> K1 embed -> K2+K4 extract -> routed
    unscramble + min-max normalise to uint8 (single:218-222) -> D2H of stego, Sc and the extracted watermark.  Three
    stages on three HIP streams (H2D, compute, D2H) with events between them and `nbuf` buffer sets in flight, so both
    DMA directions are busy continuously.  Never `value`: the contract's number is device-resident.

    Which three streams: HIP binds a stream to one of its few hardware queues when the stream is first used, and with
    some bindings an H2D copy does not start before the previous batch's D2H (a blit kernel on this stack) has finished -
    the three stages then run one after the other (1 715 frames/s instead of 2 455; timeline and the experiments that
    isolate it in profiles/r03_e2e_pipeline.md, tools/e2e_probe.py, tools/e2e_variants.py).  The binding cannot be
    chosen through the API, so up to `max_stream_sets` fresh stream triples are tried on a short run and the best one
    is timed; every attempt is in the returned object."""
    nt = (H // 8) * (W // 8)
    n = H * W
    h2d, d2h = H * W, 2 * H * W + nt * 32
    roof = pcie_roof(torch, dev, F * h2d, F * d2h)
    roof_fps = min(roof["h2d_concurrent"] * 1e9 / h2d, roof["d2h_concurrent"] * 1e9 / d2h)
    roof_fps_alone = min(roof["h2d"] * 1e9 / h2d, roof["d2h"] * 1e9 / d2h)
    h_in = [torch.randint(0, 256, (F, H, W), dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
    # one packed output record per buffer set: stego [F][H][W] u8 | watermark [F][H][W] u8 | Sc [F][nt][8] f32 (as bytes)
    o_wm, o_sc, out_bytes = F * n, 2 * F * n, 2 * F * n + F * nt * 32
    h_out = [torch.empty(out_bytes, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
    d_in = [torch.empty((F, H, W), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    d_out = [torch.empty(out_bytes, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    d_st = [t[:o_wm] for t in d_out]; d_u8 = [t[o_wm:o_sc] for t in d_out]; d_sc = [t[o_sc:] for t in d_out]

    def run(P, nb):
        s_up, s_k, s_dn, ctx, route = P
        ev_up = [torch.cuda.Event() for _ in range(nbuf)]
        ev_k = [torch.cuda.Event() for _ in range(nbuf)]
        ev_in_free = [torch.cuda.Event() for _ in range(nbuf)]
        ev_dn = [torch.cuda.Event() for _ in range(nbuf)]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for b in range(nb):
            k = b % nbuf
            with torch.cuda.stream(s_up):
                if b >= nbuf:
                    s_up.wait_event(ev_in_free[k])           # the embed that read d_in[k] is done
                d_in[k].copy_(h_in[k], non_blocking=True)
                ev_up[k].record(s_up)
            with torch.cuda.stream(s_k):
                s_k.wait_event(ev_up[k])
                if b >= nbuf:
                    s_k.wait_event(ev_dn[k])                 # the outputs of set k have left the device
                ctx.embed_tiles_u8_dev(d_in[k].data_ptr(), Sw.data_ptr(), d_st[k].data_ptr(), d_sc[k].data_ptr(), None,
                                       F, H, W, W, H * W, 0, alpha, 8)
                ev_in_free[k].record(s_k)
                ctx._call("wm_extract_unscrambled_u8_dev", api._vp(d_st[k].data_ptr()), api._vp(d_sc[k].data_ptr()), api._vp(Ux.data_ptr()),
                          api._vp(Vxt.data_ptr()), api._vp(route), api._vp(d_u8[k].data_ptr()), F, H, W, W, H * W, 0, alpha, 8, 1, 1)
                ev_k[k].record(s_k)
            with torch.cuda.stream(s_dn):
                s_dn.wait_event(ev_k[k])
                # three copies of one packed record
                h_out[k][:o_wm].copy_(d_st[k], non_blocking=True)
                h_out[k][o_wm:o_sc].copy_(d_u8[k], non_blocking=True)
                h_out[k][o_sc:].copy_(d_sc[k], non_blocking=True)
                ev_dn[k].record(s_dn)
        torch.cuda.synchronize(dev)
        return nb * F / (time.perf_counter() - t0)

    def make():
        s_up, s_k, s_dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ctx = api.Context(dev.index or 0, stream=s_k.cuda_stream)
        return (s_up, s_k, s_dn, ctx, ctx.route_dev(idx))

    attempts, best, made = [], None, []
    for _ in range(max(1, int(os.environ.get("WM_E2E_STREAM_SETS", max_stream_sets)))):
        P = make(); made.append(P)
        run(P, nbuf)
        r = run(P, 2 * nbuf)
        attempts.append(r)
        if best is None or r > best[0]:
            best = (r, P)
        if r * (2 * nbuf + 1) / (2 * nbuf) >= 0.9 * roof_fps:      # a short run pays one batch of fill / drain
            break
    fps = run(best[1], batches)
    for P in made:
        P[3].check_status(); P[3].close()
    return dict(value=fps, unit="frames/s", frames_per_batch=F, batches=batches, buffer_sets=nbuf,
                stream_sets_tried_short_run_frames_per_s=attempts,
                pcie_bytes_per_frame={"h2d": h2d, "d2h": d2h},
                pcie_GBps={"h2d": fps * h2d / 1e9, "d2h": fps * d2h / 1e9},
                pcie_roof_GBps=roof,
                frames_per_s_at_pcie_roof={"copies_alone": roof_fps_alone, "both_directions_at_once": roof_fps},
                frac_of_pcie_roof={"copies_alone": fps / roof_fps_alone, "both_directions_at_once": fps / roof_fps},
                note="pinned host memory; H2D, compute and D2H streams with events, 3 buffer sets; frame in, stego + Sc + extracted "
                     "uint8 watermark (unscrambled, normalised) out; the roof is plain pinned copies of one batch's bytes timed in "
                     "this run (best of 3 stream pairs); the pipeline's stream triple is the best of the short runs listed")


def full_extract_section(torch, api, ctx, dev, frames, Sw, Ux, Vxt, stego, sigma_c, wm_out, idx, alpha, reps=10):
    """What a user of the reference's extract gets (single:203-222), device-resident: K1 embed, then sigma + rank-8
    product (k_extract_tiles) -> routed unscramble + per-plane min-max normalise -> uint8 watermark planes.  The timed
    contract step stops at the scrambled float32 estimate; this is the rest of the chain, with HIP events per stage."""
    F, H, W = frames.shape
    n = H * W
    route = ctx.route_dev(idx)
    out_u8 = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
    vp = api._vp

    def chain():
        ctx.event_record(40)
        ctx.embed_tiles_u8_dev(frames.data_ptr(), Sw.data_ptr(), stego.data_ptr(), sigma_c.data_ptr(), None, F, H, W, W, H * W, 0, alpha, 8)
        ctx.event_record(41)
        ctx._call("wm_extract_unscrambled_u8_dev", vp(stego.data_ptr()), vp(sigma_c.data_ptr()), vp(Ux.data_ptr()), vp(Vxt.data_ptr()), vp(route),
                  vp(out_u8.data_ptr()), F, H, W, W, H * W, 0, alpha, 8, 1, 1)
        ctx.event_record(42)

    chain(); torch.cuda.synchronize(dev)
    acc = np.zeros(2)
    t0 = time.perf_counter()
    for _ in range(reps):
        chain()
        torch.cuda.synchronize(dev)
        acc += [ctx.event_elapsed_ms(40 + i, 41 + i) for i in range(2)]
    wall = time.perf_counter() - t0
    acc /= reps
    # the three-step chain it replaces, stage by stage (the tail's own roofline), then the literal index pass + per-plane
    # normalise (wm_unpermute_f32_dev + wm_normalize_u8_dev) - all must give the same bytes
    ctx.extract_tiles_px_u8_dev(stego.data_ptr(), sigma_c.data_ptr(), Ux.data_ptr(), Vxt.data_ptr(), wm_out.data_ptr(), F, H, W, W, H * W, 0, alpha, 8)
    ref3 = torch.empty_like(out_u8)
    tail = 0.0
    for _ in range(reps):
        ctx.event_record(46)
        ctx._call("wm_unpermute_normalize_u8_dev", vp(wm_out.data_ptr()), vp(route), vp(ref3.data_ptr()), n, F, 1)
        ctx.event_record(47)
        torch.cuda.synchronize(dev)
        tail += ctx.event_elapsed_ms(46, 47)
    tail /= reps
    d_idx = ctx.index_dev(idx)
    tmp = torch.empty((F, H, W), dtype=torch.float32, device=dev)
    ctx.event_record(44)
    ctx._call("wm_unpermute_f32_dev", vp(wm_out.data_ptr()), vp(d_idx), vp(tmp.data_ptr()), n, F)
    lit = torch.empty_like(out_u8)
    for f in range(F):
        ctx._call("wm_normalize_u8_dev", vp(tmp[f].data_ptr()), n, 1, vp(lit[f].data_ptr()))
    ctx.event_record(45)
    torch.cuda.synchronize(dev)
    lit_ms = ctx.event_elapsed_ms(44, 45)
    same = bool(torch.equal(lit, out_u8)) and bool(torch.equal(ref3, out_u8))
    tail_bytes = 13.0 * n * F          # algorithmic: 4 (min-max) + 4 + 4 (index) + 1 per pixel
    return {"value": F / (float(acc.sum()) * 1e-3), "unit": "frames/s",
            "what": "embed + FULL extract to the uint8 watermark, device-resident, HIP events per stage: k_embed_tiles, then ONE call "
                    "wm_extract_unscrambled_u8_dev = k_extract_tiles (which also leaves its own min / max) -> k_route_p1 -> k_route_p2",
            "frames_per_launch": F, "ms_per_launch": {"embed": float(acc[0]), "extract_to_uint8_watermark": float(acc[1])},
            "us_per_frame": {"embed": float(acc[0]) * 1e3 / F, "extract_to_uint8_watermark": float(acc[1]) * 1e3 / F},
            "wall_frames_per_s_incl_host_sync": F * reps / wall,
            "unscramble_normalise_roofline": {"bound": "hbm", "achieved": tail_bytes / (tail * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                              "unit": "GB/s", "frac": tail_bytes / (tail * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "launch_ms": tail, "algorithmic_bytes_per_launch": tail_bytes, "moved_bytes_per_launch": 17.0 * n * F,
                                              "note": "the stand-alone tail wm_unpermute_normalize_u8_dev (k_minmax_planes + k_route_p1 + k_route_p2): 13 B/px "
                                                      "algorithmic (min-max 4, value 4, index 4, byte out 1), 17 B/px moved, all coalesced; inside the "
                                                      "one-call extract the min-max pass is gone (13 B/px moved)"},
            "literal_index_pass_ms_per_launch": lit_ms, "routed_equals_literal": same}


MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32 MFMA peak (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 runs at the vector rate)


def fullframe_section(a, torch, dist, api, dev, rank, world, ctx, steps, warmup, cpu_sample=True):
    """Reference semantics (tile=None): one dense SVD per plane (single:172-177, 205-218) on BASELINE config 2's
    shape, F planes per rank per step, device-resident through the wm_ref_*_dev entry points.  Returns the
    object that goes into the bench line (rank 0) - its own frames/s, MFMA roofline and CPU baseline."""
    H = a.ff_height; W = a.ff_width; F = a.ff_frames
    L = min(H, W); K = max(8, int(0.6 * L)); alpha = a.alpha
    g = torch.Generator(device=dev); g.manual_seed(4242 + rank)
    frames = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device=dev, generator=g)
    stego = torch.empty_like(frames)
    sc = torch.empty((F, L), dtype=torch.float32, device=dev)
    wm_out = torch.empty((F, H, W), dtype=torch.float32, device=dev)
    Sw = torch.zeros((L,), dtype=torch.float32, device=dev)
    Uw = torch.zeros((H, L), dtype=torch.float32, device=dev)
    Vwt = torch.zeros((L, W), dtype=torch.float32, device=dev)
    wys = None
    if rank == 0:      # rank 0 owns the watermark: DCT + dense SVD once (single:173)
        wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
        U_, S_, Vt_ = ctx.ref_svd(wys, apply_dct=True)
        Sw.copy_(torch.from_numpy(S_)); Uw.copy_(torch.from_numpy(U_)); Vwt.copy_(torch.from_numpy(Vt_))
    if world > 1:
        for t_ in (Uw, Vwt):
            dist.broadcast(t_, src=0)              # extract-side meta: once per watermark

    def step():
        if world > 1:
            dist.broadcast(Sw, src=0)              # the path's exchange step (4.3 KB in this mode)
        ctx.ref_embed_planes_u8_dev(frames.data_ptr(), Sw.data_ptr(), stego.data_ptr(), sc.data_ptr(), None,
                                    F, H, W, W, H * W, 0, alpha, K)
        sweeps = ctx.ref_last_sweeps()
        ctx.ref_extract_planes_u8_dev(stego.data_ptr(), sc.data_ptr(), Uw.data_ptr(), Vwt.data_ptr(), wm_out.data_ptr(),
                                      F, H, W, W, H * W, alpha, K)
        return sweeps

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sweeps = step()
    barrier()
    dt = time.perf_counter() - t0
    # embed alone, for the roofline of the block-Jacobi step kernels
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    ctx.ref_embed_planes_u8_dev(frames.data_ptr(), Sw.data_ptr(), stego.data_ptr(), sc.data_ptr(), None,
                                F, H, W, W, H * W, 0, alpha, K)
    torch.cuda.synchronize(dev)
    t_embed = time.perf_counter() - t1
    sweeps_e = ctx.ref_last_sweeps()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return None
    Lp = (L + 63) // 64 * 64; M = max(H, W); nbk = Lp // 32
    # matrix-core flops the embed's block Jacobi issued, as the library counted them at launch (pairs skipped as converged
    # included): flat tournament = three 32 x 32 Gram quadrants + the 64 x 64 rotation per pair and step; two-level scheme =
    # 128 x 128 Gram tiles + one (<= 384)^2 rotation per super-pair and super-step + the stage updates
    flops_issued, two_level = ctx.ref_last_flops()
    achieved = flops_issued / t_embed / 1e12
    # With uint8 planes the two-level scheme issues its Gram and rotation products on the f16 matrix pipe as three split products
    # each (hi hi + hi lo + lo hi): `achieved` stays the f32-EQUIVALENT flop count (what an f32 implementation of the same schedule
    # would issue) against the f32 matrix peak, which is the figure comparable with earlier rounds; the f16 pipe itself sees 3 x
    # those flops against its own ~2.5 PFLOP/s dense peak
    split_f16 = two_level and os.environ.get("WM_RF_HIER_F16", "3") not in ("0",)
    # SURVEY 8(d)'s algorithmic count of what the embed computes per plane: one thin SVD, 6 M N^2 + 20 N^3 (M long, N short side)
    alg_flops = F * (6.0 * M * L * L + 20.0 * float(L) ** 3)
    achieved_alg = alg_flops / t_embed / 1e12
    # the same workload at round 3's batch (16 planes per step: the flat tournament), for continuity
    v16 = None
    if F != 16 and world == 1 and not os.environ.get("WM_BENCH_NO_V16"):
        f16, s16, c16, w16 = frames[:16].contiguous(), stego[:16].contiguous(), sc[:16].contiguous(), wm_out[:16].contiguous()
        def step16():
            ctx.ref_embed_planes_u8_dev(f16.data_ptr(), Sw.data_ptr(), s16.data_ptr(), c16.data_ptr(), None, 16, H, W, W, H * W, 0, alpha, K)
            ctx.ref_extract_planes_u8_dev(s16.data_ptr(), c16.data_ptr(), Uw.data_ptr(), Vwt.data_ptr(), w16.data_ptr(), 16, H, W, W, H * W, alpha, K)
        if F >= 16:
            step16(); torch.cuda.synchronize(dev)
            t16 = time.perf_counter()
            for _ in range(2):
                step16()
            torch.cuda.synchronize(dev)
            v16 = 2 * 16 / (time.perf_counter() - t16)
    out = {"metric": "frames/sec embed+extract, full-frame (reference semantics) Y plane",
           "value": world * F * steps / dt, "unit": "frames/s", "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "dtype": "f32",
           "workload": f"full-frame (tile=None) embed+extract, {F} planes/rank/step of {W}x{H} uint8 Y, alpha={alpha}, K={K}, "
                       f"device-resident (wm_ref_*_dev)",
           "planes_per_step": F, "value_at_16_planes": v16, "two_level_block_jacobi": two_level,
           "embed_ms": t_embed * 1e3, "embed_ms_per_plane": t_embed * 1e3 / F,
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                        "achieved_algorithmic": achieved_alg, "frac_algorithmic": achieved_alg / MFMA_F32_PEAK_TFLOPS,
                        "algorithmic_flops_per_launch": alg_flops,
                        "kernel": ("two-level block Jacobi (k_hgram_h[3] + k_happly_h + k_hupdate products)" if two_level else
                                   "block-Jacobi step (k_rf_gram + k_rf_apply GEMM tiles)"),
                        "matrix_pipe": ("f16, split operands: 3 x v_mfma_f32_32x32x16_f16 per 32 x 32 x 16 tile product (f32-class accuracy)"
                                        if split_f16 else "f32: v_mfma_f32_32x32x2_f32"),
                        "f16_pipe_flops_issued_TFLOPs": (3.0 * achieved if split_f16 else None),
                        "f16_pipe_frac_of_2500_TFLOPs": (3.0 * achieved / 2500.0 if split_f16 else None),
                        "note": f"{sweeps_e} sweeps over {nbk} blocks of 32 rows; `achieved` / `frac` count the Jacobi's OWN issued matrix-core "
                                f"flops (wm_ref_last_flops) over the whole embed call (the per-pair inner solve and the finalisation GEMMs "
                                f"are in the time, not in the flops); `frac_algorithmic` prices the same time against SURVEY 8(d)'s thin-SVD "
                                f"count 6MN^2 + 20N^3 per plane - the figure to compare implementations by; `traffic`: profiles/ "
                                f"(FETCH_SIZE x 2 + WRITE_SIZE per launch of each Jacobi kernel)"}}
    if cpu_sample:
        from oracle import wm_oracle as o
        f0 = frames[0].cpu().numpy()
        t2 = time.perf_counter(); c0 = time.process_time()
        e = o.embed_plane(f0.astype(np.float32), wys, alpha, 0.6, None)
        o.extract_plane(e["stego"].astype(np.float32), e["Sc"], e["Uw"], e["Vwt"], alpha, 0.6, H, W, None)
        wall = time.perf_counter() - t2
        cores = max(1, round((time.process_time() - c0) / wall))
        st0 = stego[0].cpu().numpy(); sc0 = sc[0].cpu().numpy()
        d = np.abs(st0.astype(np.int16) - e["stego"].astype(np.int16))
        out["cpu_baseline"] = {"value": 1.0 / wall, "unit": "frames/s", "cores": int(cores), "kind": "port",
                               "sample": f"1 frame {W}x{H}: NumPy/LAPACK oracle of single:172-177,205-218 - embed "
                                         f"(incl. its watermark SVD) + extract", "host_cpus": os.cpu_count()}
        out["parity"] = {"stego_max_lsb": int(d.max()), "stego_frac_diff": float((d != 0).mean()),
                         "sigma_max_rel": float(np.max(np.abs(sc0 - e["Sc"])) / e["Sc"][0])}
    return out


def main_fullframe(a):
    """--mode fullframe: the reference-semantics workload as the line's own metric (frames shard like tile mode)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if a.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    api = importlib.import_module(PKG + ".hostapi")
    ctx = api.Context(local_rank, stream=torch.cuda.current_stream(dev).cuda_stream)
    steps = max(1, min(a.steps, 5)); warmup = max(1, min(a.warmup, 1))
    r = fullframe_section(a, torch, dist, api, dev, rank, world, ctx, steps, warmup, cpu_sample=(world == 1 and a.cpu_frames > 0))
    if rank == 0:
        out = {"metric": r["metric"], "value": r["value"], "unit": "frames/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": r["workload"], "frames_per_rank": a.ff_frames, "height": a.ff_height,
                          "width": a.ff_width, "parallelism": f"frames sharded over {world} rank(s)"},
               "roofline": r["roofline"], "embed_ms_per_plane": r["embed_ms_per_plane"]}
        for k in ("cpu_baseline", "parity", "planes_per_step", "value_at_16_planes", "two_level_block_jacobi"):
            if k in r:
                out[k] = r[k]
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


PMC_SETS = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
            ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]))


def live_pmc_section(a, F, H, W, timeout_s=150):
    """HBM traffic and VALU occupancy of k_embed_tiles measured IN THIS RUN: three child processes, each `rocprofv3
    --kernel-trace --pmc <one counter set>` round a short `bench.py --quick` of the same shape (separate passes, KiB
    units and the x2 gfx950 FETCH_SIZE correction as MI355X_MICROARCH.md's HBM section prescribes).  The children are
    started as ordinary child processes in their own session (never exec'd from this GPU-initialised process), killed
    as a group on timeout, and anything that goes wrong returns None - the caller then replays the committed profile."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    # this process already runs under a profiler (rocprofv3 / rocprof preload its tool library): no nested profiling
    if any(k.startswith(("ROCPROF", "ROCP_", "HSA_TOOLS_LIB")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="wm_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    vals, durs = {}, {}
    try:
        for tag, counters in PMC_SETS:
            cmd = [rocprof, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", os.path.join(tmp, tag), "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--quick", "--no-live-pmc", "--steps", "3", "--warmup", "1",
                   "--cpu-frames", "0", "--frames", str(F), "--height", str(H), "--width", str(W), "--alpha", str(a.alpha)]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                return None
            if rc != 0:
                return None
            acc = {}
            for f in glob.glob(os.path.join(tmp, tag, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_embed_tiles" not in r["Kernel_Name"]:
                        continue
                    acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                    acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
                    durs.setdefault(tag, {})[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            for c in counters:
                if c not in acc or not acc[c]:
                    return None
                vals[c] = sum(acc[c].values()) / len(acc[c])          # per launch
        fetch_b = vals["FETCH_SIZE"] * 1024 * 2                       # KiB; gfx950: FETCH_SIZE reports half of a coalesced read
        write_b = vals["WRITE_SIZE"] * 1024
        t = sum(durs["sq"].values()) / len(durs["sq"]) * 1e-9
        clk = vals["GRBM_GUI_ACTIVE"] / 8 / t                          # summed over the 8 XCDs
        n_waves = F * (((H // 8) * (W // 8) + 63) // 64)
        return {"traffic": fetch_b + write_b, "fetch_bytes_x2_corrected": fetch_b, "write_bytes": write_b,
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; the denominator is SIMDs x kernel time x a clock
                # ESTIMATED from GRBM_GUI_ACTIVE (sum over the 8 XCDs, reads a few % high on sub-ms dispatches): the raw ratio can
                # exceed 1 by that error, so the fraction is capped and the raw figure kept beside it
                "valu": {"busy_frac_pmc": min(1.0, vals["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * t * clk)),
                         "busy_ratio_raw_uncapped": vals["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * t * clk),
                         "insts_per_64_tile_wave": vals["SQ_INSTS_VALU"] / n_waves, "effective_clock_GHz": clk / 1e9,
                         "kernel_us_under_pmc": t * 1e6,
                         "source": "measured in this run: child `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU "
                                   "SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` pass of `bench.py --quick` at this shape"},
                "traffic_source": "measured in this run: child `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes "
                                  "(one each) of `bench.py --quick` at this shape, per k_embed_tiles launch; FETCH_SIZE x 1024 x 2 "
                                  "(gfx950 correction, an upper bound for our 4-8 B/lane reads) + WRITE_SIZE x 1024"}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_pmc_fullframe(a, timeout_s=240):
    """HBM traffic of the full-frame block Jacobi measured IN THIS RUN: two child `rocprofv3 --kernel-trace --pmc` passes
    (FETCH_SIZE, then WRITE_SIZE) round `bench.py --mode fullframe --steps 1` of the same shape; per SVD call =
    sum over the Jacobi kernels' dispatches / number of SVD calls (one k_rf_load<unsigned char> each).  KiB units and the
    x2 gfx950 FETCH_SIZE correction as MI355X_MICROARCH.md prescribes.  Anything that goes wrong returns None."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_", "HSA_TOOLS_LIB")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    csv.field_size_limit(1 << 30)
    tmp = tempfile.mkdtemp(prefix="wm_pmcff_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", WM_BENCH_NO_V16="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    jac = ("k_rf_gram", "k_rf_inner", "k_rf_apply", "k_hgram_h3", "k_hgram_h", "k_happly_h", "k_hgram", "k_hreduce", "k_hupdate", "k_happly")
    tot, per_kernel, calls = {}, {}, 0
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "fullframe", "--steps", "1", "--cpu-frames", "0",
                   "--ff-frames", str(a.ff_frames), "--ff-height", str(a.ff_height), "--ff-width", str(a.ff_width), "--alpha", str(a.alpha)]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                return None
            if rc != 0:
                return None
            acc, n_load = 0.0, 0
            for f in glob.glob(os.path.join(tmp, counter, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    if "k_rf_load<unsigned char>" in name and r["Counter_Name"] == counter:
                        n_load += 1
                    for k in jac:
                        if k + "(" in name or k + "<" in name:
                            v = float(r["Counter_Value"]) * 1024
                            acc += v
                            per_kernel.setdefault(k, {}).setdefault(counter, 0.0)
                            per_kernel[k][counter] += v
                            break
            if not n_load or not acc:
                return None
            tot[counter] = acc / n_load
            calls = n_load
            shutil.rmtree(os.path.join(tmp, counter), ignore_errors=True)
        fetch_b, write_b = 2.0 * tot["FETCH_SIZE"], tot["WRITE_SIZE"]
        return {"traffic": fetch_b + write_b, "fetch_bytes_x2_corrected": fetch_b, "write_bytes": write_b, "svd_calls_profiled": calls,
                "per_kernel_bytes_per_svd_call": {k: {"fetch_x2": 2.0 * v.get("FETCH_SIZE", 0.0) / calls, "write": v.get("WRITE_SIZE", 0.0) / calls}
                                                  for k, v in per_kernel.items()},
                "traffic_source": "measured in this run: child `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes of "
                                  "`bench.py --mode fullframe --steps 1` at this shape; bytes of the block-Jacobi kernels (k_rf_* / k_h*) per "
                                  "SVD call of `planes_per_step` planes; FETCH_SIZE x 1024 x 2 (gfx950 correction) + WRITE_SIZE x 1024"}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes
    (torch.distributed.run, rendezvous on 127.0.0.1) from this parent, which has not imported
    torch or touched HIP, relay rank 0's single JSON line and return the children's status.
    Never re-execs: a process that has initialised the GPU must not be replaced."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    for ln in r.stdout.splitlines():
        if ln not in lines[-1:]:
            print(ln, file=sys.stderr)
    if r.returncode == 0 and not lines:
        print("[bench] ranks exited 0 but printed no JSON line", file=sys.stderr)
        return 1
    if lines:
        print(lines[-1], flush=True)
    return r.returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    if a.mode == "fullframe":
        return main_fullframe(a)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    forced = a.force_collective and world == 1
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced:                                   # no launcher: a rendezvous of one on a free local port
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)

    api = importlib.import_module(PKG + ".hostapi")
    shard = importlib.import_module(PKG + ".sharding")
    stream = torch.cuda.current_stream(dev)
    ctx = api.Context(local_rank, stream=stream.cuda_stream)

    H, W, F, alpha, K = a.height, a.width, a.frames, a.alpha, 8
    nt = (H // 8) * (W // 8)
    # ---- synthetic inputs, resident in HBM before anything is timed ------------
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    frames = torch.randint(0, 256, (F, H, W), dtype=torch.uint8, device=dev, generator=g)
    stego = torch.empty_like(frames)
    sigma_c = torch.empty((F, nt, 8), dtype=torch.float32, device=dev)
    wm_out = torch.empty((F, H, W), dtype=torch.float32, device=dev)
    Sw = torch.zeros((nt, 8), dtype=torch.float32, device=dev)
    Uw = torch.zeros((nt, 8, 8), dtype=torch.float32, device=dev)
    Vwt = torch.zeros((nt, 8, 8), dtype=torch.float32, device=dev)
    wys_np = None
    if rank == 0:   # rank 0 owns the watermark: scrambled plane -> tile SVD (K3)
        wys_np = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
        wys = torch.from_numpy(wys_np).to(dev)
        ctx.svd_tiles_f32_dev(wys.data_ptr(), Uw.data_ptr(), Sw.data_ptr(), Vwt.data_ptr(), 1, H, W, W, H * W)
        torch.cuda.synchronize(dev)
    shard.broadcast_watermark([Uw, Vwt], src=0, force=forced)      # extract-side meta: once per watermark
    # once per watermark as well: fold the IDCT into the factors (Ux = D^T Uw, Vxt = Vwt D), so the
    # per-frame extract is sigma + a rank-8 product (wm_extract_tiles_px_u8_dev)
    Ux = torch.empty_like(Uw); Vxt = torch.empty_like(Vwt)
    ctx.tile_factors_to_pixel_dev(Uw.data_ptr(), Vwt.data_ptr(), Ux.data_ptr(), Vxt.data_ptr(), nt)

    # The path's exchange step: rank 0's watermark singular values reach every rank once per
    # step (RCCL broadcast).  It is double-buffered and issued asynchronously for step k+1
    # while step k's kernels run, so the collective overlaps compute instead of serialising.
    Sw_buf = [Sw, Sw.clone()]
    pending = [None, None]

    def issue_bcast(k):
        if world > 1 or forced:
            pending[k & 1] = dist.broadcast(Sw_buf[k & 1], src=0, async_op=True)

    def step(k, record=None):
        if pending[k & 1] is not None:
            pending[k & 1].wait()                    # orders the compute stream after the broadcast
            pending[k & 1] = None
        sw = Sw_buf[k & 1]
        issue_bcast(k + 1)
        if record is not None:
            ctx.event_record(record)
        ctx.embed_tiles_u8_dev(frames.data_ptr(), sw.data_ptr(), stego.data_ptr(), sigma_c.data_ptr(), None,
                               F, H, W, W, H * W, 0, alpha, K)
        if record is not None:
            ctx.event_record(record + 1)
        ctx.extract_tiles_px_u8_dev(stego.data_ptr(), sigma_c.data_ptr(), Ux.data_ptr(), Vxt.data_ptr(),
                                    wm_out.data_ptr(), F, H, W, W, H * W, 0, alpha, K)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    issue_bcast(0)
    for k in range(a.warmup):
        step(k)
    barrier()
    n_ev = min(a.steps, 31)                           # HIP-event pairs around the embed launches
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k, 2 * k if k < n_ev else None)
    barrier()
    dt = time.perf_counter() - t0
    for w_ in pending:                                # the broadcast issued for the step after the last one
        if w_ is not None:
            w_.wait()
    if world > 1 or forced:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ctx.check_status()
    bcast_ms = None
    if forced:                                       # the collective on its own: 10 synchronous broadcasts of Sw, then the scalar gather
        torch.cuda.synchronize(dev)
        tb = time.perf_counter()
        for _ in range(10):
            dist.broadcast(Sw_buf[0], src=0)
        torch.cuda.synchronize(dev)
        bcast_ms = (time.perf_counter() - tb) / 10 * 1e3
        shard.gather_scalars(dt)

    embed_ms = [ctx.event_elapsed_ms(2 * k, 2 * k + 1) for k in range(n_ev)]
    embed_ms_avg = float(np.mean(embed_ms)) if embed_ms else float("nan")

    if rank == 0:
        value = world * F * a.steps / dt
        alg_bytes = 3.0 * H * W * F                                  # 3 B per pixel per plane
        achieved = alg_bytes / (embed_ms_avg * 1e-3) / 1e9
        valu = None
        traffic = None    # HBM bytes per embed launch from the committed PMC passes (profiles/), scaled by frames
        traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_embed_latest.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if (j.get("H"), j.get("W")) == (H, W):
                    traffic = j["hbm_bytes_per_launch_at_bench_shape"] * F / j["frames_per_launch"]
                    traffic_source = (f"REPLAYED, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      f"command committed as {j.get('source')} (measured {j.get('measured', 'round 2')} at "
                                      f"{j.get('effective_clock_GHz', 0):.2f} GHz), scaled to {F} frames per launch")
                if "valu_busy_fraction" in j:      # what actually binds this kernel (PMC pass, profiles/)
                    valu = {"busy_frac_pmc": j["valu_busy_fraction"], "insts_per_64_tile_wave": j["valu_insts_per_wave"],
                            "effective_clock_GHz": j["effective_clock_GHz"],
                            "source": f"REPLAYED from {j.get('source')} (rocprofv3 --pmc SQ_* pass, measured {j.get('measured', 'round 2')}), "
                                      f"not measured in this run"}
            except Exception:
                traffic = None
        live = None
        if world == 1 and not a.quick and not a.no_live_pmc:
            t_l = time.perf_counter()
            live = live_pmc_section(a, F, H, W)
            if live is not None:
                traffic, traffic_source, valu = live["traffic"], live["traffic_source"], live["valu"]
                valu["fetch_bytes_x2_corrected"] = live["fetch_bytes_x2_corrected"]; valu["write_bytes"] = live["write_bytes"]
                valu["collection_s"] = time.perf_counter() - t_l
            elif traffic_source is not None:
                traffic_source = "live rocprofv3 passes failed or timed out in this run; " + traffic_source
        out = {
            "metric": "frames/sec embed+extract @4K Y-channel",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"tile-mode (8x8) embed+extract, {F} frames/rank/step of {W}x{H} uint8 Y, alpha={alpha}, K=8"
                                   + (", watermark-sigma RCCL broadcast per step (async, double-buffered)" if (world > 1 or forced)
                                      else ", single rank: no broadcast"),
                       "frames_per_rank": F, "frames_per_launch": F, "height": H, "width": W, "alpha": alpha,
                       "parallelism": f"frames sharded over {world} rank(s)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "k_embed_tiles (+ its fallback pass)", "launch_ms": embed_ms_avg,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "valu": valu,
                         "note": "path is FP32-VALU-bound (~6.5e3 VALU instructions per 64-tile wave against 192 B "
                                 "per tile, VALU busy 0.99): the HBM fraction cannot approach 1, see DESIGN.md 3.3"},
        }
        if world == 1 and a.cpu_frames > 0:
            n = min(a.cpu_frames, F)
            cb, par = cpu_baseline(frames[:n].cpu().numpy(), wys_np, alpha,
                                   stego[:n].cpu().numpy(),
                                   sigma_c[:n].cpu().numpy().reshape(n, H // 8, W // 8, 8),
                                   wm_out[:n].cpu().numpy())
            out["cpu_baseline"] = cb
            out["parity"] = par
            if not a.quick:
                out["cpu_baseline_pool"] = cpu_baseline_pool(frames[:16].cpu().numpy(), alpha)
                out["cpu_baseline_reference_semantics"] = cpu_reference_semantics(frames[0].cpu().numpy(), wys_np, alpha)
        skip = set(os.environ.get("WM_BENCH_SKIP", "").split(","))         # development: leave sections out
        if world == 1 and not a.no_fullframe and not a.quick and (H, W) == (2160, 3840) and "fullframe" not in skip:
            # after the timed tile-mode region (value / ms_per_step above are untouched): the reference's own
            # full-frame semantics on BASELINE config 2's shape, with its own roofline and CPU baseline
            out["fullframe"] = fullframe_section(a, torch, dist, api, dev, rank, world, ctx, 2, 1,
                                                 cpu_sample=a.cpu_frames > 0)
            if not a.no_live_pmc:
                torch.cuda.synchronize(dev)
                live_ff = live_pmc_fullframe(a)
                if live_ff:
                    rf = out["fullframe"]["roofline"]
                    rf["traffic"] = live_ff["traffic"]
                    rf["traffic_detail"] = {k: v for k, v in live_ff.items() if k != "traffic"}
                    rf["traffic_over_plane_bytes"] = live_ff["traffic"] / (a.ff_frames * 4.0 * a.ff_height * a.ff_width)
        if world == 1 and not a.quick:
            hg = importlib.import_module(PKG + ".hostglue")
            idx = hg.permutation_index(H, W, hg.derive_key("bench", bytes(8)))       # single:62-69, host (NumPy PCG64)
            if "full_extract" not in skip:
                out["full_extract"] = full_extract_section(torch, api, ctx, dev, frames, Sw, Ux, Vxt, stego, sigma_c, wm_out, idx, alpha)
            if "end_to_end" not in skip:
                out["end_to_end"] = end_to_end_section(torch, api, dev, H, W, alpha, Sw, Ux, Vxt, idx)
        if forced:
            out["force_collective"] = {"backend": a.backend, "world_size": 1, "bcast_ms_per_step": bcast_ms,
                                       "bcast_bytes": int(Sw.numel() * 4),
                                       "note": "--force-collective: process group of one rank, the per-step broadcast issued asynchronously "
                                               "inside every timed step as at N > 1; bcast_ms_per_step = the collective alone, synchronous"}
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
