import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_world():
    """2 junction maps, 8 scenarios, 16 agent slots (built once per session: the grid index is pure Python)."""
    from torchdriveenv_amd.synth import synthetic_world

    return synthetic_world(n_scn=8, A=16, seed=0, n_maps=2)


@pytest.fixture(scope="session")
def small_world_a8():
    from torchdriveenv_amd.synth import synthetic_world

    return synthetic_world(n_scn=8, A=8, seed=1, n_maps=2)


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reward_golden.json")) as f:
        return json.load(f)
