"""The N > 1 path on a GPU: two ranks (both on device 0 - the test box has one GPU; the process
group is gloo, which stages device tensors through the host) run the bench's per-rank flow with the
real kernels: rank 0 decomposes the watermark, the decomposition is broadcast as DEVICE tensors,
each rank embeds and extracts its frame range through the device-pointer API on torch's stream.
The two shares must tile the batch and equal the single-process result bit for bit."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG_NAME

pytestmark = pytest.mark.gpu

H, W, N, ALPHA = 64, 96, 7, 0.15


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _frames():
    return np.random.default_rng(77).integers(0, 256, (N, H, W), dtype=np.uint8)


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    api = importlib.import_module(PKG_NAME + ".hostapi")
    shm = importlib.import_module(PKG_NAME + ".sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = api.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    nt = (H // 8) * (W // 8)
    Sw = torch.zeros((nt, 8), dtype=torch.float32, device=dev)
    Uw = torch.zeros((nt, 8, 8), dtype=torch.float32, device=dev); Vwt = torch.zeros_like(Uw)
    if rank == 0:
        wys = torch.from_numpy(np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)).to(dev)
        ctx.svd_tiles_f32_dev(wys.data_ptr(), Uw.data_ptr(), Sw.data_ptr(), Vwt.data_ptr(), 1, H, W, W, H * W)
        torch.cuda.synchronize(dev)
    shm.broadcast_watermark([Sw, Uw, Vwt], src=0)
    lo, hi = shm.frame_range(rank, world, N)
    n = hi - lo
    frames = torch.from_numpy(_frames()[lo:hi]).to(dev)
    stego = torch.empty_like(frames)
    sc = torch.empty((n, nt, 8), dtype=torch.float32, device=dev)
    wm = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    ctx.embed_tiles_u8_dev(frames.data_ptr(), Sw.data_ptr(), stego.data_ptr(), sc.data_ptr(), None, n, H, W, W, H * W, 0, ALPHA, 8)
    ctx.extract_tiles_u8_dev(stego.data_ptr(), sc.data_ptr(), Uw.data_ptr(), Vwt.data_ptr(), wm.data_ptr(), n, H, W, W, H * W, 0, ALPHA, 8)
    torch.cuda.synchronize(dev)
    ctx.check_status()
    np.savez(os.path.join(tmp, f"r{rank}.npz"), lo=lo, hi=hi, stego=stego.cpu().numpy(), sc=sc.cpu().numpy(),
             wm=wm.cpu().numpy(), sw=Sw.cpu().numpy())
    ctx.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_single_process_run(tmp_path, gpu_ctx):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(2)]
    assert (int(r[0]["lo"]), int(r[0]["hi"]), int(r[1]["lo"]), int(r[1]["hi"])) == (0, 3, 3, 7)
    assert np.array_equal(r[0]["sw"], r[1]["sw"]) and r[0]["sw"].max() > 0          # rank 1 received the broadcast
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    U, S, Vt = gpu_ctx.svd_tiles(wys)
    st, sc, _ = gpu_ctx.embed_tiles(_frames(), S, ALPHA)
    wm = gpu_ctx.extract_tiles(st, sc, U, Vt, ALPHA)
    assert np.array_equal(np.concatenate([r[0]["stego"], r[1]["stego"]]), st)
    assert np.array_equal(np.concatenate([r[0]["sc"], r[1]["sc"]]).reshape(sc.shape), sc)
    assert np.array_equal(np.concatenate([r[0]["wm"], r[1]["wm"]]), wm)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent (which never touches the GPU)
    starts two ranks as child processes, rank 0's single JSON line comes back with n_gpus == 2.
    gloo + --same-device because the test box has one GPU; the driver's node runs the same code over RCCL."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
                        "--height", "64", "--width", "96", "--frames", "4", "--steps", "3", "--warmup", "1", "--quick"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["value"] > 0 and j["config"]["frames_per_rank"] == 4
    assert "roofline" in j and j["roofline"]["achieved"] > 0


def test_bench_fullframe_mode_shards_frames_too():
    """`bench.py --mode fullframe --gpus 2` (reference semantics, one dense SVD per plane): the same launcher, the
    same per-rank frame batches and watermark broadcast as the tile mode - two gloo ranks on the one GPU here."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "fullframe", "--gpus", "2", "--backend", "gloo",
                        "--same-device", "--ff-height", "96", "--ff-width", "160", "--ff-frames", "2", "--steps", "2",
                        "--warmup", "1", "--cpu-frames", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["frames_per_rank"] == 2 and j["roofline"]["bound"] == "mfma"
