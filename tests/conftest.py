"""Test configuration: makes the drop-in modules importable by bare name, exactly as the
reference is used (``PYTHONPATH=src``, reference ``run_tests.sh:1``), and registers the
``gpu`` marker.  GPU tests are skipped when no device is visible."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "transcoder-sae-circuits_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
