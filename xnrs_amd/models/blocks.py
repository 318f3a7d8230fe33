"""The building blocks of the xnrs scoring path as thin nn.Module shells over HIP kernels.

Every class keeps the NAME, CONSTRUCTOR SIGNATURE, PARAMETER NAMES / REGISTRATION ORDER and forward
tensor contract of its counterpart in the reference's `xnrs.models.components` (cited per class), so
checkpoints (`state_dict`) and callers are interchangeable; none of them computes anything in torch --
`forward` is one call into `xnrs_amd.ops` (libxnrs_hip.so).  CPU tensors raise XnrsHipError.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import hip, ops


# ------------------------------------------------------------------------------------------- poolers / attention
class MaskedMean(nn.Module):
    """Reference: xnrs/models/components/layers.py:19-37.  (B,N,D),(B,N,1) -> (B,1,D) = sum(x*m)/(sum(m)+1e-8)."""

    def forward(self, x: torch.Tensor, m: torch.Tensor):
        return ops.masked_mean(x, m)


class AdditiveAttention(nn.Module):
    """Reference: layers.py:40-69.  fc1: Linear(in, hidden), fc2: Linear(hidden, 1); weights are
    exp(fc2(tanh(fc1 x))) * m / (sum + 1e-8) -- un-stabilised exp, exactly as the reference."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, 1)

    def forward(self, x: torch.Tensor, m: torch.Tensor = None, return_weights: bool = False):
        return ops.additive_attention(x, m, self, return_weights)


class MultiHeadAttention(nn.Module):
    """Reference: layers.py:105-156.  Four Linear(d_model, d_model) registered as q, v, k, (dropout), out;
    query-ROW mask with -1e9 fill; Dropout(p) on the probabilities in train mode."""

    def __init__(self, n_heads, d_model, dropout=0.1, scaled=True):
        super().__init__()
        self.h, self.d_model, self.d_k, self.scaled = n_heads, d_model, d_model // n_heads, scaled
        for name in ("q_linear", "v_linear", "k_linear"):  # the reference's registration order
            setattr(self, name, nn.Linear(d_model, d_model))
        self.dropout = nn.Dropout(dropout)
        self.out = nn.Linear(d_model, d_model)

    def forward(self, x: torch.Tensor, m: torch.Tensor):
        return ops.multi_head_attention(x, m, self)


# ------------------------------------------------------------------------------------------- encoders
def _mlp_head(n_in: int, n_out: int, activation: nn.Module, bias: bool) -> nn.Sequential:
    hip.head_activation(activation)  # nn.ReLU (reference default), nn.Tanh, nn.Identity; anything else raises here
    return nn.Sequential(nn.Linear(n_in, n_out, bias=bias), activation, nn.Linear(n_out, n_out, bias=bias))


class _Tower(nn.Module):
    """What TextEncoder and UserEncoder share: dummy_param (so `.device` works for parameter-free towers),
    input dropout, optional self-attention, pooler, optional MLP head -- in the reference's registration
    order so state_dict keys line up."""

    def _setup(self, p_dropout: float, att: Optional[nn.Module], pooler: nn.Module):
        self.dummy_param = nn.Parameter(torch.zeros(1))
        self.dropout = nn.Dropout(p=p_dropout)
        self.att = att
        self.pooler = pooler

    def _to_own_device(self, *tensors):
        dev = self.dummy_param.device
        return tuple(t.to(dev) for t in tensors)


class TextEncoder(_Tower):
    """Reference: xnrs/models/components/news_encoding.py:8-60.
    forward((x:(B,N,S,D), m:(B,N,S,1))) -> (y:(B,N,out_features), news mask:(B,N,1)).

    Extension: forward_ids(table_x, table_m, ids) encodes news gathered by row id from a device-resident
    token table (the gather is fused into the first GEMM's load)."""

    def __init__(self, pooler: nn.Module, p_dropout: float, out_features: int, in_features: Optional[int] = 768,
                 head: bool = True, activation: nn.Module = nn.ReLU(), att: Optional[nn.Module] = None, bias: bool = True):
        super().__init__()
        self._setup(p_dropout, att, pooler)
        if head:
            assert in_features is not None, 'in_features is required if head is True'
            self.head = _mlp_head(in_features, out_features, activation, bias)
        self.out_dim = out_features

    # Opt-in: encode only the news that have at least one unmasked token.  An all-masked news (an
    # empty history slot, dataset.py:82-85) pools to exactly 0 whatever its x is -- every pooling weight is
    # exp(e)*0 (layers.py:62-65) -- so its vector is the constant head(0), and the row-mask quirk cannot reach
    # it (all of its query rows are masked).  The non-empty news are gathered by row id inside the first GEMM's
    # load phase, one empty row rides along to produce the constant, and the vectors are scattered back.
    # Identical results (forward and gradients); data-dependent work, so benchmarks report it separately (it costs
    # one host sync).
    skip_empty: bool = False
    # Opt-in (inference, additive pooler, 0/1 masks): additionally skip the masked TOKEN rows wherever they cannot
    # reach the output (query projection, attention rows, output projection, fc1, pooling); K and V are still
    # projected for every token because the reference masks query rows only (ops.text_encoder_forward_unpadded).
    # Masks must be binary.  On supported shapes this runs the DEVICE-compacted entry point, which never synchronises and so
    # cannot raise for a fractional mask: its outputs are NaN and hip.STATUS_NONBINARY_MASK is set in the sticky status
    # word (hip.check_status() at the caller's next sync point raises); the host-compacted path (other shapes,
    # XNRS_COMPACT_ON_DEVICE=0) raises ValueError per call, the dense path computes exp(e)*m like the reference.  Validate a
    # dataset once with ops.check_binary_mask / NewsStore.validate_masks().
    unpadded: bool = False

    def _encoder_fn(self):
        if self.unpadded and not torch.is_grad_enabled() and hasattr(self.pooler, "fc1"):
            return ops.text_encoder_unpadded
        return ops.text_encoder

    def forward(self, inpt: tuple):
        x, m = self._to_own_device(*inpt)
        b, n, s, d = x.shape
        xf, mf = self.dropout(x.reshape(b * n, s, d)), m.reshape(b * n, s, 1)
        encode = self._encoder_fn()
        if encode is ops.text_encoder_unpadded and ops.COMPACT_ON_DEVICE and ops.compact_supported(s, d, self.att, self.pooler) \
                and not (self.att is not None and self.att.training and self.att.dropout.p > 0):
            # the device-compacted padding-free encoder drops empty news by itself: no host-side list, no sync
            y, hm = encode(xf, mf, self)
            return y.reshape(b, n, self.out_dim), hm.reshape(b, n, 1)
        if self.skip_empty and not (torch.is_grad_enabled() and xf.requires_grad):  # no input gradient through a gather
            live = mf.reshape(b * n, s).ne(0).any(dim=1)
            idx = live.nonzero().squeeze(1)
            k = idx.numel()
            if 0 < k < b * n:
                first_empty = (~live).nonzero()[:1].squeeze(1)
                ids = torch.cat([idx, first_empty]).to(torch.int32)
                yv, hv = encode(xf, mf, self, ids=ids)
                # scatter back by indexing (differentiable: in training the empty representative collects the
                # gradient of every empty slot, e.g. towards the head biases)
                pos = torch.full((b * n,), k, dtype=torch.int64, device=xf.device)
                pos[idx] = torch.arange(k, device=xf.device)
                return yv[pos].reshape(b, n, self.out_dim), hv[pos].reshape(b, n, 1)
        y, hm = encode(xf, mf, self)
        return y.reshape(b, n, self.out_dim), hm.reshape(b, n, 1)

    def forward_ids(self, table_x: torch.Tensor, table_m: torch.Tensor, ids: torch.Tensor, dedup: bool = False):
        """dedup=True encodes every distinct row once and scatters the vectors back (SURVEY.md section 8f rank 1:
        "unique-news dedup per step"); it changes the algorithmic work, so benchmarks report it separately."""
        if self.training and self.dropout.p > 0:
            # forward() applies the input dropout (news_encoding.py:51); a gathered table row cannot be dropped out
            # in the GEMM load, so training through the id path with p_dropout > 0 would silently train another model
            raise hip.XnrsHipError("TextEncoder.forward_ids: input dropout (p_dropout > 0) is not applied on the id-gather "
                                   "path; train with p_dropout = 0 (every shipped config) or through forward() on "
                                   "NewsStore.gather(ids)")
        b, n = ids.shape
        flat = ids.reshape(-1)
        encode = self._encoder_fn()
        if dedup:
            uniq, inv = torch.unique(flat, return_inverse=True)  # index bookkeeping only
            y, hm = encode(table_x, table_m, self, ids=uniq)
            y, hm = y[inv], hm[inv]
        else:
            y, hm = encode(table_x, table_m, self, ids=flat)
        return y.reshape(b, n, self.out_dim), hm.reshape(b, n, 1)


class UserEncoder(_Tower):
    """Reference: xnrs/models/components/user_encoding.py:6-81 (head = Linear(E,E)-ReLU-Linear(E,E) as
    committed at :26-34; `out_dim` and `add_features` are accepted and ignored exactly like the reference).
    forward((x:(B,N,E), m:(B,N,1))) -> (B,1,E) [, pooling weights (B,N,1)]."""

    def __init__(self, pooler: nn.Module, p_dropout: float, emb_dim: Optional[int] = None, out_dim: Optional[int] = None,
                 att: Optional[nn.Module] = None, head: bool = False, activation: nn.Module = nn.ReLU(), bias: bool = True):
        super().__init__()
        self._setup(p_dropout, att, pooler)
        if head:
            assert emb_dim is not None
            self.head = _mlp_head(emb_dim, emb_dim, activation, bias)

    def forward(self, inpt: tuple, add_features: Optional[dict] = None, return_weights: bool = False):
        x, m = self._to_own_device(*inpt)
        return ops.user_encoder(self.dropout(x), m, self, return_weights)


class DotScoring(nn.Module):
    """Reference: xnrs/models/components/scoring.py:6-23.  u:(B,1,D), c:(B,N,D) -> (B,N,1)."""

    def __init__(self, normalize: bool = False):
        super().__init__()
        self.normalize = normalize

    def forward(self, u: torch.Tensor, c: torch.Tensor):
        return ops.dot_scoring(u, c, self.normalize)


# ------------------------------------------------------------------------------------------- bi-encoder shell
class ParentRec(nn.Module):
    """Reference: xnrs/models/components/parent.py:8-81: news tower over history and candidates, user tower
    over the history vectors, scorer."""

    def __init__(self, news_encoder: nn.Module, user_encoder: nn.Module, rec_model: nn.Module, text_feature: str = 'title_emb'):
        super().__init__()
        self.news_encoder, self.user_encoder, self.rec_model = news_encoder, user_encoder, rec_model
        self.text_feature = text_feature

    def _score(self, h, hm, c, add_user_feats, return_embeddings):
        u = self.user_encoder((h, hm), add_user_feats)
        r = self.rec_model(u, c)
        return (r, u, c) if return_embeddings else r

    def _forward(self, history: Tuple[torch.Tensor], candidates: Tuple[torch.Tensor],
                 add_user_feats: Optional[Tuple[torch.Tensor]] = None, return_embeddings: bool = False):
        # (Round 4 measured the candidates' encode on a second HIP stream beside the user tower, forward and -- through
        # autograd's stream bookkeeping -- backward: bitwise the same step, but 8.28 instead of 8.17 ms for NRMS and 1.29
        # instead of 1.20 ms for StandardRec: the event traffic costs more than the latency it hides.  One stream.)
        if self._one_news_call(history, candidates):
            # serving a few impressions: history and candidates through ONE news-encoder call (a news vector depends on its
            # own rows only: bit for bit the vectors of two calls) -- half the launches of a latency-bound request
            H = history[0].shape[1]
            x = torch.cat([history[0], candidates[0]], dim=1)
            m = torch.cat([history[1], candidates[1]], dim=1)
            v, vm = self.news_encoder((x, m))
            return self._score(v[:, :H].contiguous(), vm[:, :H].contiguous(), v[:, H:].contiguous(), add_user_feats, return_embeddings)
        h, hm = self.news_encoder(history)
        c, _ = self.news_encoder(candidates)
        return self._score(h, hm, c, add_user_feats, return_embeddings)

    #: token bytes (history + candidates) up to which an inference forward concatenates the two sides into one encoder call
    ONE_CALL_MAX_BYTES = 16 << 20

    def _one_news_call(self, history, candidates) -> bool:
        if torch.is_grad_enabled() or self.training or self.ONE_CALL_MAX_BYTES <= 0:
            return False
        hx, cx = history[0], candidates[0]
        if not (isinstance(hx, torch.Tensor) and isinstance(cx, torch.Tensor)) or hx.dim() != 4 or cx.dim() != 4:
            return False
        if hx.shape[0] != cx.shape[0] or hx.shape[2:] != cx.shape[2:] or hx.device != cx.device or hx.dtype != cx.dtype:
            return False
        return (hx.numel() + cx.numel()) * hx.element_size() <= self.ONE_CALL_MAX_BYTES

    def forward(self, batch: dict, return_embeddings: bool = False):
        uf = batch['user_features']
        return self._forward(uf['history'][self.text_feature], batch['candidate_features'][self.text_feature],
                             uf['other'], return_embeddings)

    def get_user_embeddings(self, batch: dict) -> torch.Tensor:
        """(B, E) user embedding from the history alone -- what the contrastive loss consumes."""
        history = batch['user_features']['history'][self.text_feature]
        h, hm = self.news_encoder(tuple(history) if isinstance(history, list) else history)
        return self.user_encoder((h, hm)).squeeze(1)

    def forward_ids(self, table_x: torch.Tensor, table_m: torch.Tensor, hist_ids: torch.Tensor, cand_ids: torch.Tensor,
                    return_embeddings: bool = False, dedup: bool = False):
        """Extension: impressions given as row ids into a device-resident token table (row 0 / any all-zero
        row = the empty history slot).  dedup=True encodes each distinct news of the step once."""
        if dedup:
            H = hist_ids.shape[1]
            both = torch.cat([hist_ids, cand_ids], dim=1)
            v, vm = self.news_encoder.forward_ids(table_x, table_m, both, dedup=True)
            h, hm, c = v[:, :H], vm[:, :H], v[:, H:]
        elif not torch.is_grad_enabled():
            # inference: one encoder call over the ids of both sides (no token copy -- the rows are gathered inside the first
            # GEMM's load either way; bit for bit the vectors of two calls)
            H = hist_ids.shape[1]
            v, vm = self.news_encoder.forward_ids(table_x, table_m, torch.cat([hist_ids, cand_ids], dim=1))
            h, hm, c = v[:, :H], vm[:, :H], v[:, H:]
        else:
            h, hm = self.news_encoder.forward_ids(table_x, table_m, hist_ids)
            c, _ = self.news_encoder.forward_ids(table_x, table_m, cand_ids)
        return self._score(h.contiguous(), hm.contiguous(), c.contiguous(), None, return_embeddings)

    # ---- the per-model hooks of the device-resident data path (xnrs_amd.data / xnrs_amd.evaluation): every model on
    # the path answers the same three questions, whatever its news tower consumes
    def encode_news_ids(self, store, ids: torch.Tensor, dedup: bool = False):
        """News vectors of table rows `ids:(B,N)` of a NewsStore -> (vecs:(B,N,E), news mask:(B,N,1))."""
        tx, tm = store.text(self.text_feature)
        return self.news_encoder.forward_ids(tx, tm, ids, dedup=dedup)

    def encode_user(self, h: torch.Tensor, hm: torch.Tensor) -> torch.Tensor:
        """(B,H,E) history vectors + (B,H,1) mask -> (B,1,E) user vector."""
        return self.user_encoder((h, hm), None)

    def forward_store(self, store, hist_ids: torch.Tensor, cand_ids: torch.Tensor, return_embeddings: bool = False,
                      dedup: bool = False):
        """forward() with the batch given as table rows of a NewsStore (what DeviceBatcher hands out)."""
        tx, tm = store.text(self.text_feature)
        return self.forward_ids(tx, tm, hist_ids, cand_ids, return_embeddings=return_embeddings, dedup=dedup)
