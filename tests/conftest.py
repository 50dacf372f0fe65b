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
    import numpy as np

    import chess2rt_amd as c2

    class CheckedContext(c2.Context):
        """Frames with opts.count_rays = 1 come from the counting kernel instances, frames without from
        the production instances (same source, the bookkeeping compiled out).  Tests ask for counted
        frames; every such request also renders the production frame and insists on the same bits, so
        that what is compared with the oracle is what a caller gets."""

        def renderFrame(self, cam, opts, stop_flag=None):
            if not opts.count_rays:
                return super().renderFrame(cam, opts, stop_flag)
            plain_opts = type(opts).from_buffer_copy(opts)
            plain_opts.count_rays = 0
            plain = super().renderFrame(cam, plain_opts, stop_flag)
            counted = super().renderFrame(cam, opts, stop_flag)
            assert np.array_equal(plain.view(np.uint32), counted.view(np.uint32)), "production and counting instances differ"
            return counted

    ctx = CheckedContext(0)  # raises without a GPU: there is no CPU fallback
    yield ctx
    ctx.close()
