"""The multi-GPU path of bench.py on the one GPU a test box has: a one-rank RCCL process group
in the same process as librt_hip.so (both must share one HIP runtime), the renderer writing
into a torch-owned accumulator on torch's current stream, then the same collectives bench.py
issues.  With one rank every collective is the identity, so the image must not change.
(World sizes above one are covered on CPU with gloo, tests/test_distributed_cpu.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from raytracing_amd import hip, host
from raytracing_amd.distributed import new_accumulator, pass_range

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_one_rank_rccl_reduce_keeps_the_image(cornell):
    w, h, md, tile, spp = 320, 184, 5, (64, 46), 6
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(free_port())
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        torch.cuda.set_device(0)
        r = hip.HipRenderer(w, h, md, tile, device=0)
        r.upload_scene(cornell.desc)
        first, count = pass_range(spp, dist.get_world_size(), dist.get_rank())
        assert (first, count) == (0, spp)
        r.upload_seeds(host.seed_schedule(first, count, r.tile_count), first)
        cam = cornell.camera(w, h)
        accum = new_accumulator(h, w, device)
        stream = torch.cuda.current_stream().cuda_stream
        r.render(cam, first, count, accum_dev=accum.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        before = accum.cpu().numpy().copy()
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        times = torch.tensor([1.25], dtype=torch.float64, device=device)
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        assert float(times.item()) == 1.25
        assert np.array_equal(accum.cpu().numpy(), before, equal_nan=True)
        # the renderer's own accumulator path gives the same image
        r.clear()
        r.render(cam, first, count)
        own, passes = r.read_accum()
        assert passes == spp
        assert np.array_equal(own, before, equal_nan=True)
        r.close()
    finally:
        dist.destroy_process_group()
