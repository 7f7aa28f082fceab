"""rt_hip_reduce: the multi-GPU sum of the accumulators under the C ABI.  On a one-GPU box the
renderers share the GPU (the same-GPU branch adds them with a kernel); the RCCL branch is
driven through bin/Raytracing in tests/test_gpu_entry_point.py and, across real GPUs, by
`bench.py --gpus N` / `Raytracing --devices`."""
import numpy as np
import pytest

from raytracing_amd import hip, host
from raytracing_amd.distributed import pass_range

pytestmark = pytest.mark.gpu


def test_reduce_sums_pass_ranges_onto_the_first_renderer(hyperion):
    w, h, md, tile, spp, world = 640, 360, 5, (64, 36), 10, 3
    cam = hyperion.camera(w, h)
    seeds = host.seed_schedule(0, spp, 100)
    single = hip.HipRenderer(w, h, md, tile)
    single.upload_scene(hyperion.desc)
    single.upload_seeds(seeds)
    single.clear()
    single.render(cam, 0, spp)
    want, _ = single.read_accum()
    single.close()
    ranks = []
    try:
        for rank in range(world):
            r = hip.HipRenderer(w, h, md, tile, device=0)
            r.upload_scene(hyperion.desc)
            r.upload_seeds(seeds)
            r.clear()
            first, count = pass_range(spp, world, rank)   # 4 + 3 + 3
            r.render(cam, first, count)
            ranks.append(r)
        hip.reduce(ranks)
        got, passes = ranks[0].read_accum()
        assert passes == spp
        assert np.allclose(got[..., :3], want[..., :3], rtol=2e-6, atol=1e-6, equal_nan=True)
        assert (got[..., 3] == 1.0).all()
        for r in ranks[1:]:
            img, p = r.read_accum()
            assert p == 0 and (img[..., :3] == 0).all() and (img[..., 3] == 1.0).all()
        # accumulation goes on after a reduce
        ranks[1].render(cam, 0, 1)
        hip.reduce(ranks)
        assert ranks[0].read_accum()[1] == spp + 1
    finally:
        for r in ranks:
            r.close()


def test_reduce_argument_errors(cornell):
    a = hip.HipRenderer(64, 64, 3, (32, 32))
    b = hip.HipRenderer(128, 64, 3, (32, 32))
    try:
        with pytest.raises(RuntimeError, match="differ in size"):
            hip.reduce([a, b])
        with pytest.raises(RuntimeError, match="listed twice"):
            hip.reduce([a, a])
        hip.reduce([a])   # one renderer: nothing to add
    finally:
        a.close()
        b.close()
