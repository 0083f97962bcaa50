"""RCCL itself, at world size 1, on the GPU box: the collectives the N > 1 runs issue - the
broadcast of the watermark's singular values as a DEVICE tensor (4.1 MB at 4K in tile mode) and the scalar gather of
the report - over `backend="nccl"` (= RCCL on ROCm), so that the driver's 8-GPU run is not the library's first run.
The ranks are child processes started before they touch the GPU (never an exec from a GPU-initialised process)."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG_NAME

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, port, tmp):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    shm = importlib.import_module(PKG_NAME + ".sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=1, device_id=dev)
    nt = (2160 // 8) * (3840 // 8)
    sw = torch.arange(nt * 8, dtype=torch.float32, device=dev).reshape(nt, 8)        # 4.1 MB: the 4K tile-mode sigma_w
    ref = sw.clone()
    out = shm.broadcast_watermark([sw], src=0, force=True)
    work = dist.broadcast(sw, src=0, async_op=True)                                   # the bench's asynchronous form
    work.wait()
    torch.cuda.synchronize(dev)
    g = shm.gather_scalars(3.25)
    ok = bool(torch.equal(out[0], ref)) and bool(torch.equal(sw, ref))
    json.dump({"ok": ok, "backend": dist.get_backend(), "gathered": g.tolist(), "bytes": int(sw.numel() * 4)},
              open(os.path.join(tmp, "rccl.json"), "w"))
    dist.destroy_process_group()


def test_rccl_broadcast_and_gather_at_world_size_one(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = json.load(open(tmp_path / "rccl.json"))
    assert r["ok"] and r["backend"] == "nccl" and r["gathered"] == [3.25] and r["bytes"] == 4147200


def test_bench_force_collective_reports_the_broadcast():
    """`bench.py --force-collective` at N = 1: the process group is RCCL with one rank, the per-step broadcast runs inside
    the timed steps, and the line carries `force_collective.bcast_ms_per_step`."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-collective", "--height", "256", "--width", "384",
                        "--frames", "4", "--steps", "3", "--warmup", "1", "--quick"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    fc = j["force_collective"]
    assert j["n_gpus"] == 1 and fc["backend"] == "nccl" and fc["bcast_ms_per_step"] > 0 and fc["bcast_bytes"] == (256 // 8) * (384 // 8) * 32
    assert "broadcast per step" in j["config"]["workload"]
