"""pytest configuration: markers and import paths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


# PyTorch ships its own HIP runtime.  A process that lets librelp_engine.so bring in the system one first and
# imports torch afterwards ends up with two runtimes, and torch then reports "No HIP GPUs are available".  The
# tests that exchange device buffers through torch (shard tests) need torch's to be the one: import it first.
try:
    import torch  # noqa: F401,E402
except ImportError:
    pass
