"""Randomised parity: scenes with unusual parameters (radii of zero, negative or huge size, cameras inside spheres, roughness beyond
1, refraction indices below 1 or negative, emission of 1e30, flipped or zero normals, zero bounces ...) rendered by the library and
by the oracle must agree bit for bit in primary ids, accumulation and image (tools/fuzz_parity.py makes and compares them)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7, 8])
def test_random_scenes_bit_exact(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "60", str(seed)], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 of 60 cases differ" in out.stdout
