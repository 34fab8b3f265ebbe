import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the soundness tests corrupt messages (ZKCNN_MODE_TAMPER) and resident witness values (zkcnn_session_poke) of the PRODUCT library on purpose:
# both hooks refuse to work unless the process opts in
os.environ.setdefault("ZKCNN_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """On a GPU box, bring PyTorch's HIP runtime up before libzkcnn_hip.so creates its first context -- the order bench.py uses.
    (torch ships its own libamdhip64; initialising it after another copy of the runtime has been active in the process was seen to
    report `No HIP GPUs are available`.) Nothing happens on a CPU-only box."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:       # noqa: BLE001 - the CPU suite does not need torch's GPU side
        pass
    yield


@pytest.fixture(scope="session")
def built():
    """native libraries, built once per test session (hipcc cross-compiles without a GPU)"""
    import __graft_entry__
    __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    """CPU checker: oracle/liboracle.so (test infrastructure)"""
    from tests import oracle_ffi
    return oracle_ffi.load()


@pytest.fixture(scope="session")
def hip(built):
    import zkcnn_amd
    ctx = zkcnn_amd.HipContext(0)
    yield ctx
    ctx.close()
