"""Renders unusual configurations (max_depth 0 / 1 / 20, a 64x64 frame of 8x8 tiles, one pass
per call, 1920x1080) through the pipeline in several back-to-back calls and checks each image
bit for bit against the fused kernel.  usage (on a GPU box): PYTHONPATH=. python tools/robust_check.py"""
import numpy as np
from raytracing_amd import hip, host, scenes
sc = scenes.load("hyperion_rect_light")
def run(w,h,md,tile,spp,flags,calls=4):
    r = hip.HipRenderer(w,h,md,tile,accel=hip.ACCEL_REBUILT,flags=flags)
    try:
        r.upload_scene(sc.desc)
        r.upload_seeds(host.seed_schedule(0, spp*calls, r.tile_count))
        r.clear()
        cam = sc.camera(w,h)
        for c in range(calls):
            r.render(cam, c*spp, spp)
        img, passes = r.read_accum()
        return img, passes, r.ray_binning(), r.trace_refill()
    finally:
        r.close()
for (w,h,md,tile,spp) in [(1280,720,0,(128,72),8),(1280,720,1,(128,72),8),(1280,720,20,(128,72),8),(64,64,7,(8,8),4),(1280,720,7,(128,72),1),(1920,1080,7,(192,108),4)]:
    a = run(w,h,md,tile,spp,0)
    b = run(w,h,md,tile,spp,hip.FLAG_FUSED)
    same = np.array_equal(a[0], b[0], equal_nan=True)
    print((w,h,md,spp), 'passes', a[1], 'same as fused', same, 'binning', a[2][0], 'refill', a[3][0], hex(a[3][3]))
    assert same
print("ok")
