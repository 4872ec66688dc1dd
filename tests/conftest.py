import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """The C-ABI library and the oracle are built in-tree (hipcc cross-compiles without a GPU)."""
    from mujoco_mpc_amd.build import build_native
    from oracle import pyoracle
    build_native()
    pyoracle.build()


@pytest.fixture(scope="session")
def cartpole():
    from mujoco_mpc_amd.task import load_task
    return load_task("Cartpole")


@pytest.fixture(scope="session")
def particle():
    from mujoco_mpc_amd.task import load_task
    return load_task("Particle")


@pytest.fixture(scope="session")
def particle_copy():
    from mujoco_mpc_amd.task import load_task
    return load_task("ParticleCopy")


def random_nodes(seed, N, P, nu, scale=0.6):
    rng = np.random.default_rng(seed)
    return np.clip(rng.normal(0, scale, (N, P, nu)), -1, 1)
