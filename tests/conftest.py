import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once, as __graft_entry__.build() does.
    # hipcc cross-compiles without a GPU; if it is absent the product's import error stays as loud as it is.
    lib = os.path.join(ROOT, "supersdr_amd", "libssdr.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "supersdr_amd", "csrc"), "ARCH=gfx950"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def twin():
    import twinlib
    return twinlib.load()
