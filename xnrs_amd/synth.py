"""Seeded synthetic MIND-shaped inputs and weights.

Two generators:

* ``numpy`` PCG64 (version-stable bit streams) for everything that a golden fixture depends on:
  the fixture stores only the *outputs* of the reference, inputs/weights are regenerated from the
  seed on every machine.
* ``torch`` generator on the target device for bulk benchmark inputs (never part of a fixture).

Layout follows the reference's dataset (xnrs/data/dataset.py:63-109): per impression a history
``(H,S,D)`` + mask ``(H,S,1)`` and candidates ``(C,S,D)`` + mask ``(C,S,1)``; padded history slots are
all-zero x and m (dataset.py:82-85); pad *token* positions keep non-zero x (transformer outputs at
pad positions are non-zero, xnrs/data/utils.py:58-66).
"""
from __future__ import annotations

from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def rng_for(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def fill_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, int_keys: Iterable[str] = ()) -> Dict[str, torch.Tensor]:
    """Deterministic weights for a state_dict given only its key->shape map.

    Keys are visited in sorted order so the result does not depend on registration order.  Values
    are U(-b, b) with b = 1/sqrt(fan_in) (the nn.Linear default range) so activations look like a
    freshly initialised model; 1-d tensors (biases, dummy_param) use b = 1/sqrt(len) capped at 0.5.
    """
    rng = rng_for(seed)
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if k.endswith("dummy_param"):
            out[k] = torch.zeros(shp, dtype=torch.float32)
            continue
        if "embedder" in k:  # nn.Embedding tables: N(0,1) like torch's default
            out[k] = torch.from_numpy(rng.standard_normal(shp).astype(np.float32))
            continue
        fan_in = shp[-1] if len(shp) > 1 else max(shp[0], 4)
        b = min(1.0 / np.sqrt(fan_in), 0.5)
        out[k] = torch.from_numpy(rng.uniform(-b, b, size=shp).astype(np.float32))
    return out


def token_block(rng: np.random.Generator, B: int, N: int, S: int, D: int, min_len: int = 1,
                n_valid: np.ndarray | None = None, full_pad_prob: float = 0.0):
    """One (x:(B,N,S,D), m:(B,N,S,1)) pair.

    n_valid[b] = number of real news slots of impression b (trailing slots are all-zero x and m).
    """
    x = rng.standard_normal((B, N, S, D)).astype(np.float32)
    L = rng.integers(min_len, S + 1, size=(B, N))
    m = (np.arange(S)[None, None, :] < L[:, :, None]).astype(np.float32)
    if full_pad_prob > 0:
        drop = rng.random((B, N)) < full_pad_prob
        m[drop] = 0.0
    if n_valid is not None:
        slot = np.arange(N)[None, :] >= n_valid[:, None]
        x[slot] = 0.0
        m[slot] = 0.0
    return torch.from_numpy(x), torch.from_numpy(m[..., None].copy())


def make_batch(seed: int, B: int, H: int, C: int, S: int, D: int, min_len: int = 1,
               abstract: bool = False, n_categories: int = 0, n_subcategories: int = 0,
               ragged_history: bool = True) -> dict:
    """A full reference-style batch dict (xnrs/data/dataset.py:67-158, caum.py:188-199)."""
    rng = rng_for(seed)
    n_hist = rng.integers(1, H + 1, size=(B,)) if ragged_history else np.full((B,), H)
    hx, hm = token_block(rng, B, H, S, D, min_len, n_valid=n_hist)
    cx, cm = token_block(rng, B, C, S, D, min_len)
    hist = {"title_emb": (hx, hm)}
    cand = {"title_emb": (cx, cm)}
    if abstract:
        hist["abstract_emb"] = token_block(rng, B, H, S, D, min_len, n_valid=n_hist)
        cand["abstract_emb"] = token_block(rng, B, C, S, D, min_len)
    if n_categories:
        hc = rng.integers(1, n_categories + 1, size=(B, H)).astype(np.int32)
        hc[np.arange(H)[None, :] >= n_hist[:, None]] = 0
        hist["category_index"] = torch.from_numpy(hc)
        cand["category_index"] = torch.from_numpy(rng.integers(1, n_categories + 1, size=(B, C)).astype(np.int32))
    if n_subcategories:
        hs = rng.integers(1, n_subcategories + 1, size=(B, H)).astype(np.int32)
        hs[np.arange(H)[None, :] >= n_hist[:, None]] = 0
        hist["subcategory_index"] = torch.from_numpy(hs)
        cand["subcategory_index"] = torch.from_numpy(rng.integers(1, n_subcategories + 1, size=(B, C)).astype(np.int32))
    targets = np.zeros((B, C, 1), dtype=np.float32)
    targets[:, 0, 0] = 1.0  # dataset.py:147  [1,0,0,0,0]
    themes = [f"theme{int(t)}" for t in rng.integers(0, 6, size=(B,))]
    return {
        "user_features": {"history": hist, "other": {}},
        "candidate_features": cand,
        "targets": torch.from_numpy(targets),
        "main_theme": themes,
    }


def batch_to(batch, device):
    """Recursive .to(device) for the nested batch dict (tuples of tensors included)."""
    if isinstance(batch, torch.Tensor):
        return batch.to(device)
    if isinstance(batch, dict):
        return {k: batch_to(v, device) for k, v in batch.items()}
    if isinstance(batch, tuple):
        return tuple(batch_to(v, device) for v in batch)
    return batch


def device_tokens(gen: torch.Generator, n_news: int, S: int, D: int, device, min_len: int = 5,
                  zero_tail: int = 0):
    """Bulk benchmark inputs generated on ``device``: x:(n_news,S,D) N(0,1), m:(n_news,S,1) 0/1.

    ``zero_tail`` trailing news are fully padded (all-zero x and m) like padded history slots."""
    x = torch.randn((n_news, S, D), generator=gen, device=device, dtype=torch.float32)
    L = torch.randint(min(min_len, S), S + 1, (n_news, 1), generator=gen, device=device)
    m = (torch.arange(S, device=device)[None, :] < L).to(torch.float32)
    if zero_tail:
        x[n_news - zero_tail:] = 0
        m[n_news - zero_tail:] = 0
    return x, m.unsqueeze(-1)


def click_world(n_news=300, n_sess=400, S=6, D=32, n_topics=4, seed=0):
    """News carry a topic direction in their tokens; a user clicks news of its own topic."""
    rng = np.random.default_rng(seed)
    topics = rng.standard_normal((n_topics, D)).astype(np.float32) * 2.0
    news_topic = rng.integers(0, n_topics, size=n_news)
    x = np.zeros((n_news + 1, S, D), dtype=np.float32)
    m = np.zeros((n_news + 1, S), dtype=np.float32)
    for i in range(n_news):
        L = rng.integers(2, S + 1)
        x[i + 1] = rng.standard_normal((S, D)) * 0.5 + topics[news_topic[i]]
        m[i + 1, :L] = 1
    from .data import Behaviors, NewsStore
    store = NewsStore(torch.from_numpy(x), torch.from_numpy(m), list(range(n_news)))
    by_topic = [np.where(news_topic == t)[0] + 1 for t in range(n_topics)]
    others = [np.where(news_topic != t)[0] + 1 for t in range(n_topics)]
    sessions = []
    for s in range(n_sess):
        t = int(rng.integers(0, n_topics))
        sessions.append(dict(history=rng.choice(by_topic[t], size=int(rng.integers(2, 9))).tolist(),
                             positives=rng.choice(by_topic[t], size=1).tolist(),
                             negatives=rng.choice(others[t], size=int(rng.integers(4, 12))).tolist(),
                             main_theme=f"topic{t}"))
    store.index = {i + 1: i + 1 for i in range(n_news)}  # rows are used directly as ids here
    return store, Behaviors.from_sessions(sessions, store)


def model_cfg(c: dict) -> dict:
    """The flat YAML keys make_model reads (xnrs/models/make_model.py:17-18, nrms.py:12-41, naml.py:12-59) with the
    shipped configs' defaults, at the shape given by c = {model, E, bias, h, D, H, S}."""
    return dict(
        model=c["model"], scoring="dot", total_emb_dim=c["E"], title_emb_dim=c["E"], bias=c["bias"],
        n_heads=c["h"], d_backbone=c["D"], p_dropout=0.0, cat_emb_dim=16, sub_emb_dim=16,
        n_categories=19, n_subcategories=300, catg_features=[], text_features=["title_emb"],
        user_features=[], add_features=[], hist_len=c["H"], seq_len=c["S"],
    )
