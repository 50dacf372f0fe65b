import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # build products are git-ignored; (re)build them when they are missing
    libs = [os.path.join(ROOT, "chess2rt_amd", "libc2rt.so"), os.path.join(ROOT, "oracle", "libc2rt_oracle.so")]
    if not all(os.path.exists(p) for p in libs):
        subprocess.check_call(["make", "-j%d" % min(8, os.cpu_count() or 1), "all"], cwd=ROOT)


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
