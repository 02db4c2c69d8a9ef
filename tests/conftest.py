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


def pytest_terminal_summary(terminalreporter):
    """one line per session: which checker the reference-pinned tests ran against (tests/test_hip_parity.py:
    _reference_stepper — the REAL reference compiled from /root/reference, or the C oracle where that library is absent)"""
    mod = sys.modules.get("test_hip_parity")
    uses = getattr(mod, "CHECKER_USES", None) if mod else None
    if uses and (uses["reference"] or uses["oracle"]):
        terminalreporter.write_line("checker: reference (libtds_ref.so) handed to %d tests, oracle (tds_oracle.c) to %d"
                                    % (uses["reference"], uses["oracle"]))


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
