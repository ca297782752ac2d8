import os
import sys
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime initialises: see pfpp_hip/__init__.py

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "puzzlefusion-plusplus_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return {k: v for k, v in np.load(GOLDEN / f"{name}.npz").items()}

    return load


@pytest.fixture(scope="session")
def oracle_lib():
    """builds oracle/libpfpp_oracle.so (gcc) if needed"""
    from oracle import build

    return build.build()


@pytest.fixture(scope="session")
def hip_lib():
    """builds libpfpp_hip.so (hipcc cross-compiles without a GPU) if needed"""
    from pfpp_hip import build

    return build.build()


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def weights_sd():
    from oracle import weights

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = {"vqvae": weights.vqvae_state_dict, "denoiser": weights.denoiser_state_dict,
                           "verifier": weights.verifier_state_dict}[name]()
        return cache[name]

    return get
