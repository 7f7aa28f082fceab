"""The C-ABI libraries load and export every function their headers declare (no GPU needed:
nothing is called that touches a device)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

INCLUDE = os.path.join(ROOT, "include")


def declared_functions(header, prefix):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_rt_hip_exports_match_header():
    import torch  # noqa: F401  (loads the ROCm runtime torch ships, as the GPU tests do)
    from raytracing_amd import hip
    names = declared_functions("rt_hip.h", "rt_hip_")
    assert sorted(hip.EXPORTS) == names
    lib = hip.lib()
    for name in names:
        assert getattr(lib, name) is not None


def test_rt_host_exports_match_header():
    from raytracing_amd import host
    names = declared_functions("rt_host.h", "rt_host_")
    assert len(names) >= 10
    lib = host.lib()
    for name in names:
        assert getattr(lib, name) is not None


def test_struct_sizes_match_the_reference_layouts():
    from raytracing_amd import ctypes_defs as T
    # bvh_linear_node 40, glsl_mesh 12, glsl_instance 20, vertex 32, glsl_triangle 96,
    # material 96, medium 32, light 48, camera push-constant words 48
    sizes = {T.BvhNode: 40, T.Mesh: 12, T.Instance: 20, T.Vertex: 32, T.Triangle: 96,
             T.Material: 96, T.Medium: 32, T.Light: 48, T.Camera: 48, T.Mat4: 64}
    for cls, size in sizes.items():
        assert ctypes.sizeof(cls) == size


def test_hip_path_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from raytracing_amd import hip
    with pytest.raises(RuntimeError):
        hip.HipRenderer(64, 64, 3, (32, 32))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "raytracing_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M) or \
                        "liboracle" in text or "oracle.h" in text:
                    offenders.append(os.path.join(dirpath, f))
    assert offenders == []
