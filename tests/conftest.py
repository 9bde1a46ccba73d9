import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_terminal_summary(terminalreporter):
    """fp32 parity actually achieved by this session's GPU comparisons (tests/parity_log.py)."""
    try:
        import parity_log
    except ImportError:
        return
    if parity_log.RECORDS:
        for line in parity_log.summary_lines():
            terminalreporter.write_line(line)
        parity_log.dump(ROOT)
