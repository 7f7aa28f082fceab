"""tools/compare_images.py: the image-compare step of the headless workflow (raw dumps of the
`Raytracing` binary, oracle fixtures, screenshots)."""
import json
import os
import struct
import subprocess
import sys

import numpy as np

from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "compare_images.py")


def write_raw(path, accum, passes):
    h, w, _ = accum.shape
    with open(path, "wb") as f:
        f.write(b"RTACCUM1" + struct.pack("<III", w, h, passes))
        f.write(np.ascontiguousarray(accum, dtype=np.float32).tobytes())


def run(*args):
    out = subprocess.run([sys.executable, TOOL] + list(args), capture_output=True, text=True,
                         timeout=120)
    return out.returncode, json.loads(out.stdout) if out.stdout.strip() else None, out.stderr


def test_raw_dump_against_fixture_and_limits(tmp_path):
    fixture = os.path.join(GOLDEN, "oracle_cornell_box.npz")
    z = np.load(fixture)
    spp = int(z["params"][4])
    same = str(tmp_path / "same.bin")
    write_raw(same, z["accum"], spp)
    code, report, err = run(same, fixture, "--max-bad", "0", "--max-rmse", "0")
    assert code == 0, err
    assert report["radiance"]["outside_tolerance"] == 0.0 and report["radiance"]["max_abs_error"] == 0.0
    assert report["display"]["outside_tolerance"] == 0.0
    # 1 % brighter: every lit value leaves the 1e-3 tolerance, the ratio says by how much
    brighter = str(tmp_path / "brighter.bin")
    write_raw(brighter, z["accum"] * np.float32(1.01), spp)
    code, report, _ = run(brighter, fixture, "--max-bad", "0.01")
    assert code == 1
    assert report["radiance"]["outside_tolerance"] > 0.3
    assert abs(report["radiance"]["mean_ratio"] - 1.01) < 1e-4
    assert abs(report["radiance"]["relative_rmse"] - 0.01) < 1e-4
    # NaN pixels are counted and left out of the error measures
    holed = z["accum"].copy()
    holed[3, 5, 0] = np.nan
    write_raw(str(tmp_path / "holed.bin"), holed, spp)
    code, report, _ = run(str(tmp_path / "holed.bin"), fixture, "--max-bad", "0")
    assert code == 0 and report["radiance"]["nan_pixels"] == 1


def test_fixture_against_the_reference_screenshot():
    """Different sizes and only display values on one side: box-averaged to the smaller grid."""
    code, report, err = run(os.path.join(GOLDEN, "oracle_hyperion_rect_light.npz"),
                            os.path.join(GOLDEN, "ref_hyperion_thumb.png"))
    assert code == 0, err
    assert report["size"] == [160, 90] and "radiance" not in report
    # a 4-pass render against a converged screenshot: noisy, but the same picture
    assert report["display"]["relative_rmse"] < 0.5
    assert 0.7 < report["display"]["mean_ratio"] < 1.3
