"""Randomised parity: scenes with unusual parameters (radii of zero, negative or huge size, cameras inside spheres, roughness beyond
1, refraction indices below 1 or negative, emission of 1e30, flipped or zero normals, zero bounces ...) rendered by the library and
by the oracle must agree bit for bit in primary ids, accumulation and image (tools/fuzz_parity.py makes and compares them)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,cases,env", [(7, 60, {}), (8, 60, {}), (103, 260, {}), (311, 120, {"FUZZ_WIDE": "1"}),
                                            (331, 120, {"FUZZ_WIDE": "1", "RT_PARK_MIN_NODES": "1"}),
                                            (907, 160, {"FUZZ_WIDE": "1", "FUZZ_MOVES": "1"}),
                                            (911, 100, {"FUZZ_WIDE": "1", "FUZZ_MOVES": "1", "RT_PARK_MIN_NODES": "1", "RT_ENTRY_DELAY": "0"})],
                         ids=["seed7", "seed8", "seed103", "wide", "wide-parking", "camera-moves", "camera-moves-parking"])
def test_random_scenes_bit_exact(seed, cases, env):
    """(Case 251 of seed 103 -- a coarse mesh, sixteen Renders in one launch -- holds a work unit whose primary rays pass the root
    box and then miss everything alike: its walks start at the end of the node array.)  The wide draw adds larger meshes, frames of
    1 x 1 to 130 x 70 pixels and frames split into bands over several contexts; with RT_PARK_MIN_NODES=1 bounce rays with
    direction.x == 0 go through the whole-chip kernels in these small trees too.  FUZZ_MOVES moves the camera between the launches of an
    accumulation (per-view data rebuilt in the middle of it; RT_ENTRY_DELAY=0 finds the entry nodes at a view's first Render)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), str(cases), str(seed)], cwd=ROOT,
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 of %d cases differ" % cases in out.stdout
