import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suite exercises the one-read kernels on batches of a few channels: no channel-count floors unless a test asks for the library's
    # (tests/test_gpu_parity.py::test_run_chain_floors...: SsdrEngine(..., chain_floors=LIBRARY_FLOORS))
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, as __graft_entry__.build() does.
    # hipcc cross-compiles without a GPU; if it is absent the product's import error stays as loud as it is.
    lib = os.path.join(ROOT, "supersdr_amd", "libssdr.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "supersdr_amd", "csrc"), "ARCH=gfx950"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session", autouse=True)
def _no_chain_floors():
    try:
        import supersdr_amd.engine as E
    except ImportError:                                   # (no library: the tests that need it say so themselves)
        yield
        return
    old, E.DEFAULT_CHAIN_FLOORS = E.DEFAULT_CHAIN_FLOORS, (0, 0)
    yield
    E.DEFAULT_CHAIN_FLOORS = old


@pytest.fixture(scope="session")
def twin():
    import twinlib
    return twinlib.load()


def pytest_sessionfinish(session, exitstatus):
    """what the audio comparisons of this session found against north_star's plain 1e-5 RMS (tests/tolerances.py: REPORT), as a file the
    next reader can find: gpurun_out/tolerance_report_gpu.json (-m gpu) / tolerance_report_cpu.json"""
    T = sys.modules.get("tolerances")
    if T is None or not T.REPORT:
        return
    expr = session.config.getoption("markexpr", "") or ""
    kind = "gpu" if ("gpu" in expr and "not gpu" not in expr) else "cpu"
    try:
        T.write_report(os.path.join(ROOT, "gpurun_out", "tolerance_report_%s.json" % kind))
    except OSError:
        pass
