"""CPU oracle for the xnrs user-news scoring hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The product path (``xnrs_amd``) never imports, calls or falls back to anything in here.

It is a *functional restatement* (plain torch CPU ops over an explicit ``state_dict``; no
``nn.Module``; nothing imported from the reference) of the arithmetic of the reference's
hot path.  Every function cites the reference ``file:line`` (relative to the reference repo
root) whose behaviour it restates.

Parity pin: ``tests/golden/make_golden.py`` imports the *real* reference modules in the build
container, loads PCG64-seeded weights into them, and stores their outputs in
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this oracle against those
vectors (bit-exact to <=1e-6).  The reference's own repo holds no tests / golden vectors for this
path, so the imported-reference fixtures are the pin.

All functions accept ``dtype=torch.float64`` inputs as well: the float64 run is used by the
tests as a "ground truth" to show which of (HIP fp32, torch-CPU fp32) is closer.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor
State = Dict[str, Tensor]


# --------------------------------------------------------------------------- helpers
def _sub(sd: State, prefix: str) -> State:
    """Sub-dict of a state_dict below ``prefix`` ('' keeps everything)."""
    if not prefix:
        return sd
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """nn.Linear: y = x W^T + b   (used at xnrs/models/components/layers.py:60,128-130,154)."""
    y = x @ w.transpose(-1, -2)
    if b is not None:
        y = y + b
    return y


def collaps_mask(m: Tensor, dim: int) -> Tensor:
    """xnrs/utils.py:74-75  clamp(sum(m, dim), 0, 1)."""
    return torch.clamp(torch.sum(m, dim=dim), 0, 1)


# --------------------------------------------------------------------------- layers
def masked_mean(x: Tensor, m: Tensor) -> Tensor:
    """xnrs/models/components/layers.py:26-37  sum(x*m)/(sum(m)+1e-8) over dim 1."""
    return torch.sum(x * m, dim=1, keepdim=True) / (torch.sum(m, dim=1, keepdim=True) + 1e-8)


def additive_attention(x: Tensor, m: Optional[Tensor], sd: State, return_weights: bool = False):
    """xnrs/models/components/layers.py:47-69.

    a = fc2(tanh(fc1(x))); a = exp(a) (NOT max-stabilised); a *= m; a /= (sum_N a + 1e-8);
    out = a^T x.  sd keys: fc1.weight (A,D), fc1.bias (A), fc2.weight (1,A), fc2.bias (1).
    """
    a = linear(torch.tanh(linear(x, sd["fc1.weight"], sd["fc1.bias"])), sd["fc2.weight"], sd["fc2.bias"])
    a = torch.exp(a)
    if m is not None:
        a = a * m
    a = a / (torch.sum(a, dim=1, keepdim=True) + 1e-8)
    out = torch.bmm(a.transpose(-1, -2), x)
    if return_weights:
        return out, a
    return out


def multi_head_attention(x: Tensor, m: Optional[Tensor], sd: State, n_heads: int, scaled: bool = True) -> Tensor:
    """xnrs/models/components/layers.py:120-156 (eval mode: the Dropout(0.1) at :148 is identity).

    NOTE the reference's mask is a *query-row* mask: rows with m==0 are filled with -1e9 for ALL
    keys (layers.py:142-144), so such a row's softmax is uniform 1/S and valid rows still attend
    over padded keys.  sd keys: {q,k,v}_linear.{weight,bias}, out.{weight,bias}.
    """
    B, S, D = x.shape
    if D % n_heads != 0:
        # the reference's .view(B, S, h, d_k) raises a RuntimeError here (layers.py:111,133)
        raise RuntimeError(f"shape '[{B}, {S}, {n_heads}, {D // n_heads}]' is invalid for input of size {B * S * D}")
    d_k = D // n_heads
    k = linear(x, sd["k_linear.weight"], sd["k_linear.bias"])
    q = linear(x, sd["q_linear.weight"], sd["q_linear.bias"])
    v = linear(x, sd["v_linear.weight"], sd["v_linear.bias"])
    k = k.view(B, S, n_heads, d_k).transpose(1, 2)
    q = q.view(B, S, n_heads, d_k).transpose(1, 2)
    v = v.view(B, S, n_heads, d_k).transpose(1, 2)
    att = torch.matmul(q, k.transpose(-2, -1))
    if scaled:
        att = att / math.sqrt(d_k)
    if m is not None:
        att = att.masked_fill(m.unsqueeze(1) == 0, -1e9)
    att = torch.softmax(att, dim=-1)
    out = torch.matmul(att, v)
    out = out.transpose(1, 2).contiguous().view(B, S, D)
    return linear(out, sd["out.weight"], sd["out.bias"])


# --------------------------------------------------------------------------- encoders
def _mlp_head(x: Tensor, sd: State) -> Tensor:
    """nn.Sequential(Linear, ReLU, Linear) -- news_encoding.py:25-31 / user_encoding.py:30-34."""
    h = torch.relu(linear(x, sd["0.weight"], sd.get("0.bias")))
    return linear(h, sd["2.weight"], sd.get("2.bias"))


def _pool(x: Tensor, m: Tensor, sd: State, return_weights: bool = False):
    """Dispatch on what the pooler owns: AdditiveAttention has fc1/fc2, MaskedMean has no params."""
    psd = _sub(sd, "pooler")
    if "fc1.weight" in psd:
        return additive_attention(x, m, psd, return_weights=return_weights)
    assert not return_weights
    return masked_mean(x, m)


def text_encoder(x: Tensor, m: Tensor, sd: State, n_heads: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """xnrs/models/components/news_encoding.py:34-60 (eval mode, dropout identity).

    x:(B,N,S,D) m:(B,N,S,1) -> (y:(B,N,E), hm:(B,N,1)).  att is applied iff the state_dict has
    ``att.*`` keys; the head iff it has ``head.*`` keys.
    """
    b, n, s, d = x.shape
    x = x.reshape(b * n, s, d)
    m2 = m.reshape(b * n, s, 1)
    if any(k.startswith("att.") for k in sd):
        assert n_heads is not None
        x = multi_head_attention(x, m2, _sub(sd, "att"), n_heads)
    x = _pool(x, m2, sd)
    if any(k.startswith("head.") for k in sd):
        x = _mlp_head(x, _sub(sd, "head"))
    out_dim = x.shape[-1]
    x = x.reshape(b, n, out_dim)
    hm = collaps_mask(m.reshape(b, n, s, 1), dim=2)
    return x, hm


def user_encoder(x: Tensor, m: Tensor, sd: State, n_heads: Optional[int] = None, return_weights: bool = False):
    """xnrs/models/components/user_encoding.py:50-81 (eval mode).  x:(B,N,E) m:(B,N,1) -> (B,1,E)."""
    if any(k.startswith("att.") for k in sd):
        assert n_heads is not None
        x = multi_head_attention(x, m, _sub(sd, "att"), n_heads)
    if return_weights:
        x, a = _pool(x, m, sd, return_weights=True)
    else:
        x = _pool(x, m, sd)
    if any(k.startswith("head.") for k in sd):
        x = _mlp_head(x, _sub(sd, "head"))
    if return_weights:
        return x, a
    return x


def dot_scoring(u: Tensor, c: Tensor, normalize: bool = False) -> Tensor:
    """xnrs/models/components/scoring.py:12-23.  u:(B,1,E) c:(B,C,E) -> (B,C,1)."""
    if normalize:
        u = u / u.norm(p=2, dim=2, keepdim=True)
        c = c / c.norm(p=2, dim=2, keepdim=True)
    return torch.bmm(c, u.transpose(-1, -2))


# --------------------------------------------------------------------------- assemblies
def parent_forward(hist: Tuple[Tensor, Tensor], cand: Tuple[Tensor, Tensor], sd: State,
                   n_heads: Optional[int] = None, return_embeddings: bool = False):
    """xnrs/models/components/parent.py:23-38  (NRMS / StandardRec / BaseRec all go through this).

    ``sd`` is the full model state_dict (keys news_encoder.*, user_encoder.*)."""
    nsd, usd = _sub(sd, "news_encoder"), _sub(sd, "user_encoder")
    h, hm = text_encoder(hist[0], hist[1], nsd, n_heads)
    c, _ = text_encoder(cand[0], cand[1], nsd, n_heads)
    u = user_encoder(h, hm, usd, n_heads)
    r = dot_scoring(u, c)
    if return_embeddings:
        return r, u, c
    return r


def parent_user_embeddings(hist: Tuple[Tensor, Tensor], sd: State, n_heads: Optional[int] = None) -> Tensor:
    """xnrs/models/components/parent.py:49-81 -> (B,E)."""
    h, hm = text_encoder(hist[0], hist[1], _sub(sd, "news_encoder"), n_heads)
    return user_encoder(h, hm, _sub(sd, "user_encoder"), n_heads).squeeze(1)


def naml_news_vectors(title, abstract, ctg, subctg, sd: State) -> Tuple[Tensor, Tensor]:
    """xnrs/models/full_models/naml.py:76-107, one side (history or candidates).

    title/abstract: ((B,N,S,D),(B,N,S,1)); ctg/subctg: (B,N) integer -> ((B,N,E), title mask (B,N,1))."""
    t, tm = text_encoder(title[0], title[1], _sub(sd, "title_encoder"))
    a, _ = text_encoder(abstract[0], abstract[1], _sub(sd, "body_encoder"))
    ce = linear(sd["cat_embedder.weight"][ctg.long()], sd["cat_fc.weight"], sd["cat_fc.bias"])
    se = linear(sd["subcat_embedder.weight"][subctg.long()], sd["subcat_fc.weight"], sd["subcat_fc.bias"])
    b, n, e = t.shape
    v = torch.cat([t, a, ce, se], dim=2).reshape(b * n, 4, e)
    v = additive_attention(v, None, _sub(sd, "feature_pooler")).reshape(b, n, e)
    return v, tm


def naml_forward(batch: dict, sd: State) -> Tensor:
    """xnrs/models/full_models/naml.py:61-112,149-160."""
    hf, cf = batch["user_features"]["history"], batch["candidate_features"]
    hist, hm = naml_news_vectors(hf["title_emb"], hf["abstract_emb"], hf["category_index"], hf["subcategory_index"], sd)
    cand, _ = naml_news_vectors(cf["title_emb"], cf["abstract_emb"], cf["category_index"], cf["subcategory_index"], sd)
    u = additive_attention(hist, hm, _sub(sd, "user_encoder"))
    return dot_scoring(u, cand)


def naml_user_embeddings(batch: dict, sd: State) -> Tensor:
    """xnrs/models/full_models/naml.py:113-147 -> (B,1,E) (NOT squeezed, :146-147)."""
    hf = batch["user_features"]["history"]
    hist, hm = naml_news_vectors(hf["title_emb"], hf["abstract_emb"], hf["category_index"], hf["subcategory_index"], sd)
    return additive_attention(hist, hm, _sub(sd, "user_encoder"))


def lstur_news_encoder(title, cat_idxs: Tensor, subcat_idxs: Optional[Tensor], sd: State) -> Tuple[Tensor, Tensor]:
    """xnrs/models/full_models/lstur.py:191-207: TextEncoder (+) category [(+) subcategory] embedding."""
    t, m = text_encoder(title[0], title[1], _sub(sd, "title_encoder"))
    emb = torch.cat([t, sd["cat_embedder.weight"][cat_idxs.long()]], dim=2)
    if subcat_idxs is not None:
        emb = torch.cat([emb, sd["subcat_embedder.weight"][subcat_idxs.long()]], dim=2)
    return emb, m


# --------------------------------------------------------------------------- loss / grad step
def mse_relu_loss(scores: Tensor, targets: Tensor) -> Tensor:
    """xnrs/training.py:379-392  mse_loss(relu(scores), targets)."""
    return torch.nn.functional.mse_loss(torch.relu(scores), targets)


def contrastive_loss(embeddings: Tensor, labels: Tensor, temperature: float) -> Tensor:
    """xnrs/training.py:433-472 -- in-batch InfoNCE with the reference's exact epsilons.

    Vectorised restatement of the per-row Python loop: rows without a positive are skipped,
    loss_i = -log( sum_{pos} exp(sim/t) / (sum_{j!=i} exp(sim/t) + 1e-12) ), mean = sum/(count+1e-8).
    """
    if embeddings.dim() > 2:
        embeddings = embeddings.reshape(embeddings.size(0), -1)
    e = torch.nn.functional.normalize(embeddings, dim=-1)
    sim = (e @ e.mT) / temperature
    B = e.size(0)
    eye = torch.eye(B, dtype=torch.bool, device=e.device)
    same = labels[:, None] == labels[None, :]
    pos = same & ~eye
    ex = torch.exp(sim)
    num = (ex * pos).sum(dim=1)
    den = (ex * (~eye)).sum(dim=1)
    has_pos = pos.any(dim=1)
    num = torch.where(has_pos, num, torch.ones_like(num))  # skipped rows: keep log() finite (grad-safe)
    li = -torch.log(num / (den + 1e-12))
    li = torch.where(has_pos, li, torch.zeros_like(li))
    count = has_pos.sum()
    return li.sum() / (count + 1e-8)


def contrastive_loss_loop(embeddings: Tensor, labels: Tensor, temperature: float) -> Tensor:
    """Literal per-row loop form of xnrs/training.py:448-472 (small B only; used to pin the
    vectorised form above)."""
    if embeddings.dim() > 2:
        embeddings = embeddings.reshape(embeddings.size(0), -1)
    e = torch.nn.functional.normalize(embeddings, dim=-1)
    sim = e @ e.mT
    B = e.size(0)
    loss = 0.0
    count = 0
    ar = torch.arange(B)
    for i in range(B):
        pos_mask = (labels == labels[i]) & (ar != i)
        pos_sims = sim[i][pos_mask] / temperature
        all_sims = sim[i][ar != i] / temperature
        if len(pos_sims) == 0:
            continue
        loss = loss + -torch.log(torch.exp(pos_sims).sum() / (torch.exp(all_sims).sum() + 1e-12))
        count += 1
    return loss / (count + 1e-8)


def train_step_loss(batch: dict, sd: State, n_heads: Optional[int], labels: Tensor,
                    temperature: float, lambda_cl: float) -> Tuple[Tensor, Tensor, Tensor]:
    """xnrs/training.py:402-431 for ParentRec models: loss_rec + lambda * loss_cl (eval-mode dropout)."""
    hist = batch["user_features"]["history"]["title_emb"]
    cand = batch["candidate_features"]["title_emb"]
    r = parent_forward(hist, cand, sd, n_heads)
    loss_rec = mse_relu_loss(r, batch["targets"])
    ue = parent_user_embeddings(hist, sd, n_heads)
    loss_cl = contrastive_loss(ue, labels, temperature)
    return loss_rec + lambda_cl * loss_cl, loss_rec, loss_cl
