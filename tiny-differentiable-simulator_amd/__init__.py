"""tiny-differentiable-simulator_amd — MI355X-native many-instance replay of the TDS
per-environment step.  Import through the repo-root alias ``tds_amd`` (the directory name
is not a valid Python identifier)."""
from .model import *  # noqa: F401,F403
from . import model  # noqa: F401

from . import hip_backend  # noqa: F401,E402
from . import ranks  # noqa: F401,E402
from . import vec_env  # noqa: F401,E402
from .vec_env import VectorizedAntEnv, VectorizedLaikagoEnv, VectorizedEnv  # noqa: F401,E402
