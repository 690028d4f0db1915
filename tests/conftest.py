import os
import sys

import pytest

# torch bundles its own copy of the HIP runtime: it has to be the first one the process loads, or a later torch.cuda init finds
# "No HIP GPUs" (two runtimes in one process) -- the in-tree library then binds to the copy that is already there
import torch  # noqa: E402,F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
