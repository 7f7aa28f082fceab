import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # A GPU test on a machine without a GPU is an error of selection, not a skip-to-green:
    # the driver selects with -m gpu / -m "not gpu"; anything else gets an explicit skip.
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def cornell():
    from raytracing_amd import scenes
    return scenes.load_snapshot("cornell_box")


@pytest.fixture(scope="session")
def hyperion():
    from raytracing_amd import scenes
    return scenes.load_snapshot("hyperion_rect_light")


@pytest.fixture(scope="session")
def mustang():
    """assets/mustang_red.json reduced to the five meshes the reference ships, under a
    substitute environment map (tools/make_fixtures.py: reduced_mustang)."""
    from raytracing_amd import scenes
    return scenes.load_snapshot("mustang_red_reduced")


@pytest.fixture(scope="session")
def staircase():
    """assets/staircase.json without the one primitive whose mesh does not ship
    (tools/make_fixtures.py: reduced_staircase): ten 1024x1024 albedo textures."""
    from raytracing_amd import scenes
    return scenes.load_snapshot("staircase_reduced")


@pytest.fixture(scope="session")
def teapot():
    """assets/teapot.json with a sphere standing in for the two teapot meshes that do not ship
    and a substitute environment map (tools/make_fixtures.py: teapot_standin)."""
    from raytracing_amd import scenes
    return scenes.load_snapshot("teapot_standin")


@pytest.fixture(scope="session")
def textured(tmp_path_factory):
    from raytracing_amd import host, scenes
    d = tmp_path_factory.mktemp("textured")
    return host.HostScene.from_json(scenes.write_textured_scene(str(d)))


@pytest.fixture(scope="session")
def instanced(tmp_path_factory):
    """40 x 40 = 1600 instances of one sphere: far beyond the reference shader's 15-entry
    candidate list."""
    from raytracing_amd import host, scenes
    d = tmp_path_factory.mktemp("instanced")
    path = scenes.write_instanced_scene(str(d), grid=40, sphere_segments=24, sphere_rings=12,
                                        resolution=(320, 180), tile=(64, 36))
    return host.HostScene.from_json(path)


@pytest.fixture(scope="session")
def synthetic_4k():
    """The generated 4K configuration: 10 000 instances of one 15 872-triangle sphere (159 M
    instanced triangles), three area lights, constant sky (raytracing_amd/scenes.py)."""
    from raytracing_amd import scenes
    return scenes.load("synthetic_4k")
