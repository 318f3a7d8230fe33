import os
import sys

# hipGraph replays of multi-kernel steps came out wrong with ROCm 7.2's graph "packet capture" path on gfx950 (stale reads
# between kernel nodes; tools/debug_graph_step.py); the runtime reads this switch when it is loaded, i.e.
# before the first `import torch`
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible and -m gpu was not forced."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
