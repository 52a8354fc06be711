import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"

KAT_KEY = 0x2B7E151628AED2A6ABF7158809CF4F3C


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Kit:
    """client + evaluation keys + CPU oracle for one parameter set"""

    def __init__(self, params, seed, key=KAT_KEY, iv=0x6BC1BEE22E409F96E93D7E117393172A):
        from oracle import oracle as orc
        from tfhe_aes_amd.client import Client

        self.params = params
        self.client = Client(1, iv, key, params=params, seed=seed)
        self.keys = self.client.server_keys()
        self.oracle = orc.Oracle(params, self.keys.ksk, self.keys.bsk, self.keys.pfpksk)
        self._engine = None

    def engine(self):
        """HIP engine through the C ABI with the keys uploaded (GPU tests only)"""
        if self._engine is None:
            from tfhe_aes_amd import _native

            self._engine = _native.Engine(self.params, device=0)
            self._engine.upload_keys(self.keys.ksk, self.keys.bsk, self.keys.pfpksk)
        return self._engine


@pytest.fixture(scope="session")
def toy():
    from tfhe_aes_amd import PARAM_TOY

    return Kit(PARAM_TOY, seed=0x70F)


@pytest.fixture(scope="session")
def opt():
    from tfhe_aes_amd import PARAM_OPT

    return Kit(PARAM_OPT, seed=0xAE50001)


@pytest.fixture(scope="session")
def golden():
    import json

    return json.loads((GOLDEN / "golden.json").read_text())


def sha(a) -> str:
    import hashlib

    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
