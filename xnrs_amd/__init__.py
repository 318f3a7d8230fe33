"""xnrs_amd -- MI355X (gfx950) native implementation of the xnrs user-news scoring hot path.

Host side: Python on PyTorch-ROCm mirroring the reference's ``xnrs.models`` module API.
Device side: hand-written HIP kernels in ``libxnrs_hip.so`` (C ABI in include/xnrs_hip.h).
"""
__version__ = "0.1.0"
