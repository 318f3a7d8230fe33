"""Integrated gradients of one score w.r.t. the history's token embeddings -- the compute loop of
Explainer.explain_score_in_batch (xnrs/explain.py:144-182; the one timing the reference publishes).

The reference walks the interpolation path a = 1/n, 2/n, ..., 1 one step at a time: a news-encoder forward over the H
history news scaled by a, the user encoder, the score against the fixed candidate vector, and torch.autograd.grad(score,
scaled tokens) -- n forward + input-gradient passes of ONE impression each, launch-latency bound on any GPU.  The steps are
independent of each other (each score depends on its own scaled copy only), so here they are the BATCH dimension: one
forward and one input-gradient pass over `steps_per_batch` scaled copies of the history, through the same HIP kernels as
the grad step (no parameter gradient is computed: autograd._wanted_inputs).  `batched=False` keeps the reference's loop
(same numbers up to the summation order of the final sums; tests/test_hip_explain.py).

Out of scope here as in DESIGN.md section 11: the tokenizer, the backbone that produces the token embeddings, plotting.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch


def _encode_candidate(model, cand_emb, cand_att, cidx):
    # explain.py:155-157: only one candidate is scored
    ce = cand_emb[:, cidx:cidx + 1, :, :]
    ca = cand_att[:, cidx:cidx + 1, :, :]
    c, _ = model.news_encoder((ce, ca))
    return c


def integrated_gradients(model, hist_emb: torch.Tensor, hist_att: torch.Tensor, cand_emb: torch.Tensor,
                         cand_att: torch.Tensor, candidate_idx: int = 0, n_steps: int = 100,
                         activation: Optional[Callable] = torch.relu, batched: bool = True,
                         steps_per_batch: int = 0) -> Dict[str, object]:
    """hist_emb (1, H, S, D), hist_att (1, H, S, 1), cand_emb (1, C, S, D), cand_att (1, C, S, 1) on the model's device.

    Returns {"attr": (H, S) token attributions, "news_attribution": (H,), "int_grads": (H, S, D), "s_true": score at a = 1,
    "s_attr": sum of the attributions} -- explain.py:167-173 (the dictionary of titles / tokens around them is the
    caller's business).  steps_per_batch: interpolation steps per pass (0 = all n_steps at once; n_steps * H * S * D * 4
    bytes of scaled tokens and as much again of gradients per pass)."""
    if hist_emb.dim() != 4 or hist_emb.size(0) != 1:
        raise ValueError("integrated_gradients explains ONE impression: hist_emb must be (1, H, S, D)")
    if n_steps < 1:
        raise ValueError("n_steps must be >= 1")
    act = activation if activation is not None else (lambda t: t)
    hist_emb = hist_emb.detach()
    hist_att = hist_att.detach()
    with torch.no_grad():
        c = _encode_candidate(model, cand_emb.detach(), cand_att.detach(), candidate_idx)  # (1, 1, E): fixed along the path
    da = 1.0 / n_steps
    # explain.py:160: torch.arange(da, 1 + da, da) -- the same fp32 values, cut to n_steps (rounding can add one)
    alphas = torch.arange(da, 1 + da, da, device=hist_emb.device)[:n_steps]
    H, S, D = hist_emb.shape[1:]
    int_grads = torch.zeros((H, S, D), dtype=hist_emb.dtype, device=hist_emb.device)
    s_true = None
    if not batched:
        for a in alphas:
            ga = (a * hist_emb).requires_grad_()
            ha, ham = model.news_encoder((ga, hist_att))
            ua = model.user_encoder.forward(inpt=(ha, ham))
            sa = act(model.rec_model(ua, c))
            (g,) = torch.autograd.grad(sa, ga)
            int_grads += g[0] * da
            s_true = sa.detach()
    else:
        per = n_steps if steps_per_batch <= 0 else min(int(steps_per_batch), n_steps)
        for lo in range(0, n_steps, per):
            al = alphas[lo:lo + per]
            nb = al.numel()
            ga = (al.view(nb, 1, 1, 1) * hist_emb).requires_grad_()     # (nb, H, S, D): step i is impression i of the batch
            att = hist_att.expand(nb, -1, -1, -1).contiguous()
            ha, ham = model.news_encoder((ga, att))
            ua = model.user_encoder.forward(inpt=(ha, ham))
            sa = act(model.rec_model(ua, c.expand(nb, -1, -1).contiguous()))
            (g,) = torch.autograd.grad(sa.sum(), ga)                     # d sa_i / d ga_j = 0 for i != j
            int_grads += g.sum(dim=0) * da
            s_true = sa.detach().reshape(nb, -1)[-1:]
    attr_full = int_grads * hist_emb[0]                                   # explain.py:169
    attr = attr_full.sum(dim=2)                                           # (H, S): explain.py:170 sums batch and feature axes
    return {"attr": attr, "news_attribution": attr.sum(dim=1), "int_grads": int_grads,
            "s_true": float(s_true.reshape(-1)[-1].item()), "s_attr": float(attr.sum().item())}
