"""Loss kernels next to the hot path: the in-batch InfoNCE "theme" contrastive loss of
ContrastiveRankingTrainer._compute_contrastive_loss (xnrs/training.py:433-472) as ONE fused HIP forward +
ONE fused backward instead of a Python loop over the batch rows."""
from __future__ import annotations

import torch

from . import hip


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, labels, temperature):
        emb = hip.dev_f32(emb, "embeddings")
        B, E = emb.shape
        labels = labels.to(device=emb.device, dtype=torch.int64).contiguous()
        l = hip.lib()
        nsaved = l.xnrs_infonce_saved_bytes(B, E)
        saved = torch.empty(nsaved, dtype=torch.uint8, device=emb.device)
        loss = torch.empty((), dtype=torch.float32, device=emb.device)
        hip.check(l.xnrs_infonce_fwd(hip.ptr(emb), hip.ptr(labels), B, E, float(temperature), hip.ptr(loss), hip.ptr(saved),
                                     nsaved, hip.stream_ptr(emb.device)), "xnrs_infonce_fwd")
        ctx.save_for_backward(labels, saved)
        ctx.shape, ctx.temperature, ctx.nsaved = (B, E), float(temperature), nsaved
        return loss

    @staticmethod
    def backward(ctx, g):
        labels, saved = ctx.saved_tensors
        B, E = ctx.shape
        g = hip.dev_f32(g.reshape(1), "upstream gradient")
        demb = torch.empty((B, E), dtype=torch.float32, device=saved.device)
        hip.check(hip.lib().xnrs_infonce_bwd(hip.ptr(labels), B, E, ctx.temperature, hip.ptr(saved), ctx.nsaved, hip.ptr(g),
                                             hip.ptr(demb), hip.stream_ptr(saved.device)), "xnrs_infonce_bwd")
        return demb, None, None


def contrastive_loss(embeddings: torch.Tensor, labels: torch.Tensor, temperature: float) -> torch.Tensor:
    """Drop-in for ContrastiveRankingTrainer._compute_contrastive_loss(embeddings:(B,E)|(B,1,E), labels:(B,)).
    (The `if embeddings.dim() > 2: view(B, -1)` of training.py:442-443 is kept for NAML's (B,1,E).)"""
    if embeddings.dim() > 2:
        embeddings = embeddings.reshape(embeddings.size(0), -1)
    return _InfoNCE.apply(embeddings, labels, temperature)
