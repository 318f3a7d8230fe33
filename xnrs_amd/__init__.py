"""xnrs_amd -- MI355X (gfx950) native implementation of the xnrs user-news scoring hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's ``xnrs.models`` module API.
Device side: hand-written HIP kernels in ``libxnrs_hip.so`` (C ABI in include/xnrs_hip.h).
"""
import os as _os

# hipGraph replays of the grad step: ROCm 7.2's graph packet-capture path returned stale data between kernel nodes on
# gfx950 (INTEGRATION.md "Capturing the step in a hipGraph").  The runtime reads the switch when it loads, so this helps
# only when xnrs_amd is imported before torch; callers that capture graphs set it themselves (bench.py, tests/conftest.py).
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

__version__ = "0.1.0"


def install(force: bool = False) -> bool:
    """Make the reference's own `from xnrs.models import make_model` (train.py:12) resolve to the HIP-backed modules
    without touching a reference file -- see xnrs_amd/mirrors.py."""
    from .mirrors import install as _install
    return _install(force)
