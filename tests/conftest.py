"""pytest configuration: markers, import paths, shared fixtures.

``-m "not gpu"`` runs on the build container's CPU (oracle vs golden fixtures, host logic, C-ABI
symbol checks, 2-rank gloo data-parallel test); ``-m gpu`` are the parity tests proper and need one
MI355X.
"""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("loco-asr_amd.synth")


@pytest.fixture(scope="session")
def state_dict(synth):
    return synth.encoder_state_dict(0)


@pytest.fixture(scope="session")
def oracle():
    import speecht5_oracle

    return speecht5_oracle


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def record_figure(name, **values):
    """Append measured parity figures (not only pass / fail) to gpurun_out/parity_figures.jsonl: the GPU box's copy comes back
    with every gpurun call, and the round's summary is committed under profiles/."""
    import json
    import time
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_figures.jsonl"), "a") as fh:
            fh.write(json.dumps({"test": name, "time": round(time.time(), 1), **values}) + "\n")
    except OSError:
        pass
