import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits(a):
    """View a float32 array as uint32 for exact comparisons."""
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def ref():
    """The reference's own objects (oracle/_ref); only exists where /root/reference does."""
    from oracle import ref as r
    if not r.available():
        pytest.skip("oracle/_ref/libref.so not built (reference tree absent)")
    return r


def default_sphere_scene(orc_mod):
    from raytracer_amd import scenes
    sc = orc_mod.Scene()
    sc.set_spheres(scenes.default_spheres())
    sc.set_materials(scenes.default_materials())
    return sc
