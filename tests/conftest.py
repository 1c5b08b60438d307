"""pytest configuration: `gpu` marker (tests that need a real MI355X), repo root on sys.path."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


def pytest_runtest_logreport(report):
    """Append every failure (test id + traceback) to $YMK_TEST_FAILURE_LOG when set: lets a rare flake in a long
    unattended loop be identified afterwards."""
    import os

    path = os.environ.get("YMK_TEST_FAILURE_LOG")
    if path and report.failed:
        with open(path, "a") as f:
            f.write(f"=== {report.nodeid} [{report.when}]\n{report.longreprtext}\n")
