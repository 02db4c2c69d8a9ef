import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = ["cartpole", "pendulum5", "ant", "laikago", "laikago_soft", "pendulum5_plane", "cartpole_plane",
          "ant_floating", "laikago_floating", "cube_floating", "laikago_floating_env",
          "pendulum5_spherical", "sphere_spherical", "humanoid_spherical", "humanoid",
          "humanoid_sph_pd", "pendulum5_sph_pd", "sphere_spherical_spring", "pendulum5_spherical_spring"]


# worlds with TWO articulated bodies: fixtures from the real reference (oracle/gen_golden.py)
TWO_BODY_MODELS = ["two_pendulums", "two_pendulums_plane", "two_pendulums_capsule_a", "two_pendulums_capsule_b"]
# ... with THREE / FOUR bodies (every pair i < j is a contact pass of its own, world.hpp:206-282)
MULTI_BODY_MODELS = TWO_BODY_MODELS + ["three_pendulums", "three_pendulums_plane", "four_pendulums"]
# ... with FLOATING bases among the bodies
FLOATING_MULTI_BODY_MODELS = ["two_cubes_floating", "pendulum_and_cube"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the product library and the C oracle once per session (no GPU needed)."""
    import __graft_entry__ as g

    g.build()
    return True


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor) — the 'relative per-step' metric of BASELINE.json with an
    absolute floor so that exact zeros do not blow up."""
    import numpy as np

    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
