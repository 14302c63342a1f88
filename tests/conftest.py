import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def wav_paths():
    names = ["noisy_100hz_sine.wav", "noisy_200hz_sine.wav", "noisy_300hz_sine.wav",
             "noisy_400hz_sine.wav", "noise.wav"]
    return [os.path.join(GOLDEN, "audio", n) for n in names]
