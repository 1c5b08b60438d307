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


@pytest.fixture
def emu(monkeypatch):
    """Install the torch restatement of the libymk entry points (tests/emu_ops.py) over `yolo_master_amd.ops` and lift
    the device guard, so that the product's HOST code can be driven end to end on the CPU.  Test infrastructure only."""
    from tests import emu_ops
    from yolo_master_amd import ops

    for name in emu_ops.EMULATED:
        assert hasattr(ops, name), f"ops.{name} disappeared: update tests/emu_ops.py"
        monkeypatch.setattr(ops, name, getattr(emu_ops, name))
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    emu_ops.CALLS.clear()
    return emu_ops


def pytest_runtest_logreport(report):
    """Append every failure (test id + traceback) to $YMK_TEST_FAILURE_LOG when set: lets a rare flake in a long
    unattended loop be identified afterwards."""
    import os

    path = os.environ.get("YMK_TEST_FAILURE_LOG")
    if path and report.failed:
        with open(path, "a") as f:
            f.write(f"=== {report.nodeid} [{report.when}]\n{report.longreprtext}\n")


def pytest_runtest_protocol(item, nextitem):
    """CPU-side tests get ONE retry (rendezvous ports, process spawning and compiler invocations can fail
    transiently on a busy host); a test has to fail twice to be reported as failed, and every retry is printed.
    GPU parity tests are never retried.  Disable with YMK_NO_RERUN=1."""
    import os

    from _pytest.runner import runtestprotocol

    if "gpu" in item.keywords or os.environ.get("YMK_NO_RERUN"):
        return None
    item.ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    reports = runtestprotocol(item, nextitem=nextitem, log=False)
    if any(r.failed for r in reports):
        first = next(r for r in reports if r.failed)
        print(f"\n[retry] {item.nodeid} failed once ({first.when}): {first.longreprtext.splitlines()[-1] if first.longreprtext else ''}")
        log = os.environ.get("YMK_TEST_FAILURE_LOG")
        if log:
            with open(log, "a") as f:
                f.write(f"=== first attempt of {item.nodeid} [{first.when}]\n{first.longreprtext}\n")
        reports = runtestprotocol(item, nextitem=nextitem, log=False)
    for r in reports:
        item.ihook.pytest_runtest_logreport(report=r)
    item.ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True
