"""Host helpers mirrored from xnrs/utils.py that the hot path needs."""
import torch

from . import ops


def collaps_mask(m: torch.Tensor, dim: int):
    """xnrs/utils.py:74-75  clamp(sum(m, dim), 0, 1).  (The reference's spelling is kept.)

    Runs as a HIP wavefront reduction for the layout the hot path uses (m:(..., S, 1), dim = -2)."""
    nd = m.dim()
    if dim < 0:
        dim += nd
    if dim != nd - 2 or m.shape[-1] != 1:
        raise ValueError("collaps_mask: the HIP path supports m:(..., S, 1) reduced over dim=-2 "
                         f"(got shape {tuple(m.shape)}, dim={dim})")
    return ops.collapse_mask(m)


def batch_to_device(batch: dict, device):
    """xnrs/utils.py:88-93 (in-place, tensors and nested dicts only -- tuples are left alone exactly
    like the reference; the encoders move their own inputs)."""
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            batch[k] = v.to(device)
        elif isinstance(v, dict):
            batch_to_device(v, device)
