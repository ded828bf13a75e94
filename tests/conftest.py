import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The dense solve's task lists are built by a worker thread and the first factorisations of a new system size go column by column until
# they are there (csrc/chol_dag.hip).  The suite wants the schedule a size has by DEFAULT from the first factorisation on -- the stall
# injection counts re-run trials, the schedule tests compare the task graph with the column launches -- so the build is synchronous
# here (children inherit it); tests/test_chol_gpu.py::test_first_factorisations_of_a_new_size_fall_back_and_agree covers the
# asynchronous default in a child of its own.
os.environ.setdefault("MAGE_CHOL_DAG_SYNC_BUILD", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; building it here would hide a missing artefact, so it must already exist."""
    from mageslam_amd import _lib
    return _lib.lib()


def pytest_sessionstart(session):
    """PyTorch-ROCm bundles its own HIP runtime, and in a process that also loads libmageslam_hip.so (system runtime) torch's
    has to initialise first (mageslam_amd/_lib.py): some GPU tests only import torch after the first handle exists."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:       # noqa: CPU-only environments
        pass
