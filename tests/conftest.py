import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ablation: A/B kernels that live only in the measurement library (round 6 prune) -- run on a GPU box with "
                                       "X2I_LIB_VARIANT=ablate python -m pytest tests -m ablation; skipped everywhere else")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        # the fp32 CPU oracle of the full-depth parity tests: torch's CPU GEMMs peak near half the physical cores on the 2-socket hosts of the pool
        # (bench.py's cpu_baseline pins the same count); the default -- every hardware thread -- ran those tests 2x slower
        torch.set_num_threads(max(1, min(64, (os.cpu_count() or 4) // 4)))
        if os.environ.get("X2I_LIB_VARIANT") != "ablate":
            need = pytest.mark.skip(reason="A/B kernels of the measurement library: run with X2I_LIB_VARIANT=ablate")
            for item in items:
                if "ablation" in item.keywords:
                    item.add_marker(need)
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords or "ablation" in item.keywords:
            item.add_marker(skip)
