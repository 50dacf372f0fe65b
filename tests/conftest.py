import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # Build products are git-ignored (they still travel to the GPU box with the working tree).  `make all`
    # every session: a no-op when libc2rt.so / the oracle libraries are newer than their sources, a rebuild
    # when a source changed or a library is missing — so the tests never run a stale binary.
    subprocess.check_call(["make", "-s", "-j%d" % min(8, os.cpu_count() or 1), "all"], cwd=ROOT)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def scenes_dir():
    return os.path.join(ROOT, "tests", "golden", "scenes")


@pytest.fixture(scope="session")
def gpu_ctx():
    import chess2rt_amd as c2

    ctx = c2.Context(0)  # raises without a GPU: there is no CPU fallback
    yield ctx
    ctx.close()
