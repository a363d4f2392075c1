import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# A few GPU tests flip the library's developer switches (GL_QKV_FUSED, GL_GEMM_WIDE) to reach kernels the default selection does
# not pick; the library reads them only with GL_DEV_SWITCHES=1 (csrc/common.h dev_env), decided at its first use in the process.
os.environ.setdefault("GL_DEV_SWITCHES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """A native engine on cuda:0 (session-wide: the arena is 6 GiB)."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from gligen_amd.build import build_native
    from gligen_amd.engine import Engine

    build_native()
    eng = Engine(0, arena_gb=6.0)
    yield eng
    eng.close()
