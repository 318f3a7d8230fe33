"""Tensor-level entry points of the HIP hot path (forward).  Each function validates its inputs,
sizes the workspace, and makes ONE call into libxnrs_hip.so on the current HIP stream.

Shapes follow the reference's module contracts; masks are fp32 0/1 with a trailing
singleton dim exactly as the reference passes them.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import hip


def _mask2d(m: Optional[torch.Tensor], rows: int, cols: int, what: str) -> Optional[torch.Tensor]:
    if m is None:
        return None
    m = hip.dev_f32(m, what)
    if m.numel() != rows * cols:
        raise RuntimeError(f"{what}: mask of shape {tuple(m.shape)} does not match ({rows}, {cols}, 1)")
    return m.reshape(rows, cols)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = hip.ACT_NONE):
    """nn.Linear forward (+ fused activation) on the fp32 MFMA GEMM.  x:(..., K) -> (..., N)."""
    if act == hip.ACT_NONE and torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (x, weight, bias)):
        from . import autograd
        return autograd.linear(x, weight, bias)
    x = hip.dev_f32(x, "linear input")
    w = hip.dev_f32(weight, "linear weight")
    b = None if bias is None else hip.dev_f32(bias, "linear bias")
    K = x.shape[-1]
    N = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x.numel() // K}x{K} and {w.shape[1]}x{N})")
    M = x.numel() // K
    y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().xnrs_linear_fwd(hip.ptr(x), None, 0, hip.ptr(w), hip.ptr(b), hip.ptr(y), M, N, K, act,
                                        hip.stream_ptr(x.device)), "xnrs_linear_fwd")
    return y


def mha_forward(x: torch.Tensor, m: Optional[torch.Tensor], att, dropout_p: float = 0.0, seed: int = 0):
    """layers.MultiHeadAttention.forward (layers.py:120-156).  x:(B,S,D), m:(B,S,1)|None -> (B,S,D)."""
    x = hip.dev_f32(x, "mha input")
    if x.dim() != 3:
        raise RuntimeError(f"MultiHeadAttention expects (B, S, D), got {tuple(x.shape)}")
    B, S, D = x.shape
    m2 = _mask2d(m, B, S, "mha mask")
    p, keep = hip.mha_params(att, dropout_p, seed)
    y = torch.empty_like(x)
    l = hip.lib()
    nbytes = l.xnrs_mha_workspace_bytes(B, S, D)
    ws = hip.workspace(x.device, nbytes)
    hip.check(l.xnrs_mha_fwd(hip.ptr(x), hip.ptr(m2), C.byref(p), hip.ptr(y), B, S, D, hip.ptr(ws), nbytes,
                             hip.stream_ptr(x.device)), "xnrs_mha_fwd")
    return y


def additive_forward(x: torch.Tensor, m: Optional[torch.Tensor], pool, return_weights: bool = False):
    """layers.AdditiveAttention.forward (layers.py:47-69).  x:(B,N,D), m:(B,N,1)|None -> (B,1,D) [,(B,N,1)]."""
    x = hip.dev_f32(x, "additive input")
    if x.dim() != 3:
        raise RuntimeError(f"AdditiveAttention expects (B, N, D), got {tuple(x.shape)}")
    B, N, D = x.shape
    m2 = _mask2d(m, B, N, "additive mask")
    p, keep = hip.additive_params(pool)
    y = torch.empty((B, 1, D), dtype=torch.float32, device=x.device)
    a = torch.empty((B, N, 1), dtype=torch.float32, device=x.device) if return_weights else None
    l = hip.lib()
    nbytes = l.xnrs_additive_workspace_bytes(B, N, D, p.hidden)
    ws = hip.workspace(x.device, nbytes)
    hip.check(l.xnrs_additive_attention_fwd(hip.ptr(x), hip.ptr(m2), C.byref(p), hip.ptr(y), hip.ptr(a), B, N, D,
                                            hip.ptr(ws), nbytes, hip.stream_ptr(x.device)),
              "xnrs_additive_attention_fwd")
    return (y, a) if return_weights else y


def masked_mean(x: torch.Tensor, m: torch.Tensor):
    """layers.MaskedMean.forward (layers.py:26-37).  x:(B,N,D), m:(B,N,1) -> (B,1,D)."""
    if torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad:
        from . import autograd
        return autograd.masked_mean(x, m)
    x = hip.dev_f32(x, "masked_mean input")
    B, N, D = x.shape
    m2 = _mask2d(m, B, N, "masked_mean mask")
    y = torch.empty((B, 1, D), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().xnrs_masked_mean_fwd(hip.ptr(x), hip.ptr(m2), hip.ptr(y), B, N, D, hip.stream_ptr(x.device)),
              "xnrs_masked_mean_fwd")
    return y


def collapse_mask(m: torch.Tensor):
    """xnrs.utils.collaps_mask(m, dim=-2) for m:(..., S, 1) -> (..., 1)  (xnrs/utils.py:74-75)."""
    m = hip.dev_f32(m, "mask")
    S = m.shape[-2]
    rows = m.numel() // S
    hm = torch.empty(m.shape[:-2] + (1,), dtype=torch.float32, device=m.device)
    hip.check(hip.lib().xnrs_collapse_mask(hip.ptr(m), hip.ptr(hm), rows, S, hip.stream_ptr(m.device)),
              "xnrs_collapse_mask")
    return hm


def _pool_args(pooler, att=None):
    """att: the attention stage in front of the pooler in an INFERENCE call (the cached folded fc1 rides along)."""
    from .models.components import layers
    if isinstance(pooler, layers.AdditiveAttention):
        p, keep = hip.additive_params(pooler, att)
        return hip.POOL_ADDITIVE, p, keep
    if isinstance(pooler, layers.MaskedMean):
        return hip.POOL_MEAN, None, []
    raise hip.XnrsHipError(f"pooler {type(pooler).__name__} has no HIP implementation "
                           "(supported: AdditiveAttention, MaskedMean)")


def text_encoder_forward(x: torch.Tensor, m: torch.Tensor, att, pooler, head, ids: Optional[torch.Tensor] = None,
                         chunk: int = 0, dropout_p: float = 0.0, seed: int = 0):
    """TextEncoder.forward core (news_encoding.py:48-59) on (n_news,S,D)/(n_news,S) -> (y:(n,E'), hm:(n,)).

    With ``ids`` (int32, (n,)), ``x``/``m`` are the device-resident news-token table and its mask and
    the rows are gathered inside the first GEMM's load phase."""
    x = hip.dev_f32(x, "text encoder input")
    n_tab, S, D = x.shape
    m2 = _mask2d(m, n_tab, S, "text encoder mask")
    if ids is not None:
        if not ids.is_cuda:
            raise hip.XnrsHipError("ids must live on the HIP device")
        ids = ids.to(torch.int32).contiguous()
        n = ids.numel()
    else:
        n = n_tab
    pool_kind, pp, keep = _pool_args(pooler, att if dropout_p == 0.0 else None)
    ap = hp = None
    if att is not None:
        ap, k2 = hip.mha_params(att, dropout_p, seed)
        keep += k2
    if head is not None:  # (behind attention + additive pooling the out-projection folds into the head's first layer)
        hp, k3 = hip.head_params(head, att if (pool_kind == hip.POOL_ADDITIVE and dropout_p == 0.0) else None)
        keep += k3
    A = pp.hidden if pp is not None else 0
    E = hp.out_features if hp is not None else D
    y = torch.empty((n, E), dtype=torch.float32, device=x.device)
    hm = torch.empty((n,), dtype=torch.float32, device=x.device)
    l = hip.lib()
    nbytes = l.xnrs_text_encoder_workspace_bytes(n, S, D, A, E, int(att is not None), pool_kind, int(head is not None),
                                                 chunk)
    ws = hip.workspace(x.device, nbytes)
    hip.check(l.xnrs_text_encoder_fwd(hip.ptr(x), hip.ptr(m2), hip.ptr(ids), n, S, D,
                                      None if ap is None else C.byref(ap), pool_kind,
                                      None if pp is None else C.byref(pp), None if hp is None else C.byref(hp),
                                      hip.ptr(y), hip.ptr(hm), chunk, hip.ptr(ws), nbytes, hip.stream_ptr(x.device)),
              "xnrs_text_encoder_fwd")
    return y, hm


def text_encoder_forward_unpadded(x: torch.Tensor, m: torch.Tensor, att, pooler, head, ids: Optional[torch.Tensor] = None,
                                  news_per_pass: int = 8192):
    """TextEncoder.forward computing only what can reach the output (include/xnrs_hip.h:
    xnrs_text_encoder_fwd_unpadded): the unmasked token rows are compacted (index bookkeeping with torch, one
    host sync for the counts), K/V are projected for every row, everything else for the live rows only.
    Inference, additive pooler, 0/1 masks.  Same signature and results as text_encoder_forward."""
    x = hip.dev_f32(x, "text encoder input")
    n_tab, S, D = x.shape
    m2 = _mask2d(m, n_tab, S, "text encoder mask")
    if ids is not None:
        if not ids.is_cuda:
            raise hip.XnrsHipError("ids must live on the HIP device")
        ids = ids.to(torch.int32).contiguous()
        live = m2[ids.long()].ne(0)                      # (n, S) mask of the gathered news
        base = ids.long().unsqueeze(1) * S               # table token row of (news, 0)
    else:
        live = m2.ne(0)
        base = None
    n = live.shape[0]
    pool_kind, pp, keep = _pool_args(pooler, att)
    if pool_kind != hip.POOL_ADDITIVE:
        raise hip.XnrsHipError("the unpadded encoder needs the additive pooler")
    bad = ((m2 != 0) & (m2 != 1)).any()
    ap = hp = None
    if att is not None:
        ap, k2 = hip.mha_params(att, 0.0, 0)
        keep += k2
    if head is not None:
        hp, k3 = hip.head_params(head, att)
        keep += k3
    A, E = pp.hidden, (hp.out_features if hp is not None else D)
    y = torch.empty((n, E), dtype=torch.float32, device=x.device)
    hm = torch.empty((n,), dtype=torch.float32, device=x.device)
    cnt = live.sum(dim=1)
    off = torch.zeros(n + 1, dtype=torch.int64, device=x.device)
    torch.cumsum(cnt, 0, out=off[1:])
    tok = torch.arange(S, device=x.device).unsqueeze(0)
    l = hip.lib()
    step = max(1, int(news_per_pass))
    bounds = list(range(0, n, step)) + [n]
    off_host = off[bounds].tolist()                       # the one host sync (also surfaces `bad`)
    if bool(bad):
        raise hip.XnrsHipError("the unpadded encoder needs a 0/1 mask")
    for c0, c1, r0, r1 in zip(bounds[:-1], bounds[1:], off_host[:-1], off_host[1:]):
        nc, nv = c1 - c0, r1 - r0
        lv = live[c0:c1]
        if ids is not None:
            rows = (base[c0:c1] + tok)[lv].to(torch.int32)             # table token rows of the live tokens
            xp, idp = x, ids[c0:c1]
        else:
            rows = (torch.arange(nc, device=x.device).unsqueeze(1) * S + tok)[lv].to(torch.int32)  # rows inside this pass
            xp, idp = x[c0:c1], None
        roff = (off[c0:c1 + 1] - r0).contiguous()
        nbytes = l.xnrs_text_encoder_unpadded_workspace_bytes(nc, nv, S, D, A, E, int(att is not None), int(head is not None))
        ws = hip.workspace(x.device, nbytes)
        hip.check(l.xnrs_text_encoder_fwd_unpadded(hip.ptr(xp), hip.ptr(idp), nc, S, D, hip.ptr(rows), hip.ptr(roff), nv,
                                                   None if ap is None else C.byref(ap), C.byref(pp),
                                                   None if hp is None else C.byref(hp), hip.ptr(y[c0:c1]), hip.ptr(hm[c0:c1]),
                                                   hip.ptr(ws), nbytes, hip.stream_ptr(x.device)),
                  "xnrs_text_encoder_fwd_unpadded")
    return y, hm


def text_encoder_forward_compact(x: torch.Tensor, m: torch.Tensor, att, pooler, head, ids: Optional[torch.Tensor] = None,
                                 chunk: int = 0):
    """The padding-free encoder with the row lists built on the DEVICE (include/xnrs_hip.h: xnrs_text_encoder_fwd_compact):
    same inputs and results as text_encoder_forward for 0/1 masks, no host sync anywhere -- the call can be captured in a
    hipGraph.  Raises XnrsHipError(code -4) for shapes / modes it does not serve (see compact_supported)."""
    x = hip.dev_f32(x, "text encoder input")
    n_tab, S, D = x.shape
    m2 = _mask2d(m, n_tab, S, "text encoder mask")
    if ids is not None:
        if not ids.is_cuda:
            raise hip.XnrsHipError("ids must live on the HIP device")
        ids = ids.to(torch.int32).contiguous()
        n = ids.numel()
    else:
        n = n_tab
    pool_kind, pp, keep = _pool_args(pooler, att)
    if pool_kind != hip.POOL_ADDITIVE:
        raise hip.XnrsHipError("the compact encoder needs the additive pooler")
    ap = hp = None
    if att is not None:
        ap, k2 = hip.mha_params(att, 0.0, 0)
        keep += k2
    if head is not None:
        hp, k3 = hip.head_params(head, att)
        keep += k3
    A, E = pp.hidden, (hp.out_features if hp is not None else D)
    y = torch.empty((n, E), dtype=torch.float32, device=x.device)
    hm = torch.empty((n,), dtype=torch.float32, device=x.device)
    hip.status_word(x.device)  # (registered once: a non-binary mask sets STATUS_NONBINARY_MASK there; hip.check_status())
    l = hip.lib()
    nbytes = l.xnrs_text_encoder_compact_workspace_bytes(n, S, D, A, E, int(att is not None), int(head is not None), chunk)
    # The default pass (262 144 / S news) needs ~3.2 GB of worst-case scratch at D = 768, and hip.workspace is grow-only per
    # stream: bound it (XNRS_COMPACT_WS_MB, default 4096) by shrinking the pass -- the scratch scales with the news per pass
    cap = int(os.environ.get("XNRS_COMPACT_WS_MB", "4096")) << 20
    if chunk == 0 and nbytes > cap:
        per_pass = max(262144 // max(S, 1), 1)
        while nbytes > cap and per_pass > 64:
            per_pass //= 2
            nbytes = l.xnrs_text_encoder_compact_workspace_bytes(n, S, D, A, E, int(att is not None), int(head is not None), per_pass)
        chunk = per_pass
    ws = hip.workspace(x.device, nbytes)
    hip.check(l.xnrs_text_encoder_fwd_compact(hip.ptr(x), hip.ptr(m2), hip.ptr(ids), n, S, D, None if ap is None else C.byref(ap),
                                              C.byref(pp), None if hp is None else C.byref(hp), hip.ptr(y), hip.ptr(hm), chunk,
                                              hip.ptr(ws), nbytes, hip.stream_ptr(x.device)), "xnrs_text_encoder_fwd_compact")
    return y, hm


def compact_supported(S: int, D: int, att, pooler) -> bool:
    """Mirror of the C entry point's preconditions that can be decided on the host without touching the data."""
    from .models.components import layers
    if not isinstance(pooler, layers.AdditiveAttention) or hip.get_gemm_mode() != 0 or D % 4 != 0 or S > 512:
        return False
    if os.environ.get("XNRS_FC1_ROWDOT", "1") == "0" or os.environ.get("XNRS_GEMM_BUF", "1") == "0":
        return False
    if att is not None:
        dk = D // att.h
        if S > 64 or dk > 64 or dk % 4 != 0 or os.environ.get("XNRS_FOLD_OUT", "1") == "0":
            return False
    return True


def user_encoder_forward(x: torch.Tensor, m: torch.Tensor, att, pooler, head, return_weights: bool = False,
                         dropout_p: float = 0.0, seed: int = 0):
    """UserEncoder.forward core (user_encoding.py:69-81).  x:(B,H,E), m:(B,H,1) -> (B,1,E) [,(B,H,1)]."""
    x = hip.dev_f32(x, "user encoder input")
    B, H, E = x.shape
    m2 = _mask2d(m, B, H, "user encoder mask")
    pool_kind, pp, keep = _pool_args(pooler, att if dropout_p == 0.0 else None)
    ap = hp = None
    if att is not None:
        ap, k2 = hip.mha_params(att, dropout_p, seed)
        keep += k2
    if head is not None:  # (behind attention + additive pooling the out-projection folds into the head's first layer)
        hp, k3 = hip.head_params(head, att if (pool_kind == hip.POOL_ADDITIVE and dropout_p == 0.0) else None)
        keep += k3
    A = pp.hidden if pp is not None else 0
    y = torch.empty((B, 1, E), dtype=torch.float32, device=x.device)
    a = torch.empty((B, H, 1), dtype=torch.float32, device=x.device) if return_weights else None
    l = hip.lib()
    nbytes = l.xnrs_user_encoder_workspace_bytes(B, H, E, A, int(att is not None), pool_kind, int(head is not None))
    ws = hip.workspace(x.device, nbytes)
    hip.check(l.xnrs_user_encoder_fwd(hip.ptr(x), hip.ptr(m2), B, H, E, None if ap is None else C.byref(ap), pool_kind,
                                      None if pp is None else C.byref(pp), None if hp is None else C.byref(hp),
                                      hip.ptr(y), hip.ptr(a), hip.ptr(ws), nbytes, hip.stream_ptr(x.device)),
              "xnrs_user_encoder_fwd")
    return (y, a) if return_weights else y


def dot_scoring(u: torch.Tensor, c: torch.Tensor, normalize: bool = False):
    """DotScoring.forward (scoring.py:12-23).  u:(B,1,E), c:(B,C,E) -> (B,C,1)."""
    if torch.is_grad_enabled() and (getattr(u, "requires_grad", False) or getattr(c, "requires_grad", False)):
        from . import autograd
        return autograd.dot_scoring(u, c, normalize)
    return dot_scoring_forward(u, c, normalize)


def dot_scoring_forward(u: torch.Tensor, c: torch.Tensor, normalize: bool = False):
    u = hip.dev_f32(u, "user vector")
    c = hip.dev_f32(c, "candidate vectors")
    B, Cn, E = c.shape
    if u.numel() != B * E:
        raise RuntimeError(f"batch1 and batch2 shapes do not match: u {tuple(u.shape)} vs c {tuple(c.shape)}")
    r = torch.empty((B, Cn, 1), dtype=torch.float32, device=c.device)
    hip.check(hip.lib().xnrs_dot_scoring_fwd(hip.ptr(u), hip.ptr(c), hip.ptr(r), B, Cn, E, int(normalize),
                                             hip.stream_ptr(c.device)), "xnrs_dot_scoring_fwd")
    return r


# ------------------------------------------------------------------------------------------------
# module-level dispatch: what the nn.Module mirrors call.  Handles train-mode attention dropout and
# routes to the autograd path when gradients are required.
def _needs_grad(*tensors_or_modules) -> bool:
    if not torch.is_grad_enabled():
        return False
    for t in tensors_or_modules:
        if t is None:
            continue
        if isinstance(t, torch.Tensor):
            if t.requires_grad:
                return True
        else:
            for p in t.parameters():
                if p.requires_grad:
                    return True
    return False


#: Optional DEVICE word (int64 tensor of one element) that the attention kernels add to every dropout seed
#: (include/xnrs_hip.h: xnrs_mha_params::seed_dev).  A grad step captured in a hipGraph bakes the host seeds into the graph;
#: a caller that increments this word INSIDE the captured step (`word.add_(1)`) gets a fresh attention-dropout draw per
#: replay, as the reference draws one per call (layers.py:148).  None (default): host seeds alone.
_DROPOUT_SEED_WORD = None


def set_dropout_seed_word(word):
    """word: None, or a one-element int64 tensor on the HIP device.  Do not change it between a forward and its backward."""
    global _DROPOUT_SEED_WORD
    if word is not None and not (isinstance(word, torch.Tensor) and word.is_cuda and word.dtype == torch.int64 and word.numel() == 1):
        raise hip.XnrsHipError("the dropout seed word is a one-element int64 tensor on the HIP device")
    _DROPOUT_SEED_WORD = word


def dropout_seed_word():
    return _DROPOUT_SEED_WORD


def _att_dropout(att):
    """(p, seed) of the attention-probability dropout (layers.py:117,148): active in train mode only.
    The seed is drawn from torch's CPU generator so torch.manual_seed() controls it."""
    if att is None or not att.training or att.dropout.p <= 0.0:
        return 0.0, 0
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    return float(att.dropout.p), seed


def _check_att(att):
    from .models.components import layers
    if att is not None and not isinstance(att, layers.MultiHeadAttention):
        raise hip.XnrsHipError(f"att module {type(att).__name__} has no HIP implementation "
                               "(supported: xnrs_amd MultiHeadAttention)")


def multi_head_attention(x, m, att):
    p, seed = _att_dropout(att)
    if _needs_grad(x, att):
        from . import autograd
        return autograd.mha(x, m, att, p, seed)
    return mha_forward(x, m, att, p, seed)


def additive_attention(x, m, pool, return_weights=False):
    if _needs_grad(x, pool):
        from . import autograd
        return autograd.additive(x, m, pool, return_weights)
    return additive_forward(x, m, pool, return_weights)


def text_encoder(x, m, enc, ids=None, chunk: int = 0):
    att, pooler, head = enc.att, enc.pooler, getattr(enc, "head", None)
    _check_att(att)
    p, seed = _att_dropout(att)
    if _needs_grad(x, enc):
        from . import autograd
        return autograd.text_encoder(x, m, enc, ids, p, seed)
    return text_encoder_forward(x, m, att, pooler, head, ids=ids, chunk=chunk, dropout_p=p, seed=seed)


def text_encoder_unpadded(x, m, enc, ids=None):
    """Inference-only variant of text_encoder that skips the padding work (text_encoder_forward_unpadded).  Falls
    back to the padded kernels -- loudly impossible cases aside -- when attention dropout is active (train mode)."""
    att, pooler, head = enc.att, enc.pooler, getattr(enc, "head", None)
    _check_att(att)
    p, _ = _att_dropout(att)
    S, D = x.shape[-2], x.shape[-1]
    dk = D // att.h if att is not None else 4
    # outside the unpadded kernels' range (S, d_k <= 64, 16-byte aligned head rows) the padded kernels give the same
    # result -- an optional speed-up never turns into an error
    supported = S <= 64 and D % 4 == 0 and (att is None or (dk <= 64 and dk % 4 == 0))
    if _needs_grad(x, enc) or p > 0.0 or not supported:
        return text_encoder(x, m, enc, ids=ids)
    if COMPACT_ON_DEVICE and compact_supported(S, D, att, pooler):
        # row lists built on the device: no host sync, capturable; all-masked news cost nothing (so skip_empty's host-side
        # compaction is not needed on top).  A device without room for its scratch falls back to the host-compacted path
        # (the same results) instead of failing: hip.release_workspaces() returns the scratch when eval and train alternate
        try:
            return text_encoder_forward_compact(x, m, att, pooler, head, ids=ids)
        except torch.OutOfMemoryError:
            hip.release_workspaces()
    return text_encoder_forward_unpadded(x, m, att, pooler, head, ids=ids)


def check_binary_mask(m: torch.Tensor, what: str = "mask") -> None:
    """Raise ValueError unless every mask value is 0 or 1 (ONE host sync: set-up code, never the step).  The padding-free
    paths require binary masks (an unmasked row is simply "live"); the host-compacted path checks this per call, the
    device-compacted one (TextEncoder.unpadded on supported shapes) cannot raise from the device -- it writes NaN outputs
    and sets hip.STATUS_NONBINARY_MASK in the sticky status word (hip.check_status()).  Call this once per dataset instead
    (NewsStore.validate_masks)."""
    bad = ((m != 0) & (m != 1)).any()
    if bool(bad.item()):
        raise ValueError(f"{what}: values other than 0 / 1 (the reference's masks are fp32 0/1, news_encoding.py:34-50)")


#: TextEncoder.unpadded compacts on the device when the shape allows (XNRS_COMPACT_ON_DEVICE=0: always the host-compacted path)
COMPACT_ON_DEVICE = os.environ.get("XNRS_COMPACT_ON_DEVICE", "1") != "0"


def user_encoder(x, m, enc, return_weights=False):
    att, pooler, head = enc.att, enc.pooler, getattr(enc, "head", None)
    _check_att(att)
    p, seed = _att_dropout(att)
    if _needs_grad(x, enc):
        from . import autograd
        return autograd.user_encoder(x, m, enc, return_weights, p, seed)
    return user_encoder_forward(x, m, att, pooler, head, return_weights, dropout_p=p, seed=seed)


def embedding_linear(idx: torch.Tensor, embedder, fc):
    """fc(embedder(idx)) (naml.py:82-86) as ONE GEMM whose A rows are gathered from the embedding
    table by index.  idx:(B,N) int -> (B,N,out_features)."""
    if _needs_grad(embedder, fc):
        from . import autograd
        return autograd.embedding_linear(idx, embedder, fc)
    tab = hip.dev_f32(embedder.weight, "embedding table")
    w = hip.dev_f32(fc.weight, "fc weight")
    b = None if fc.bias is None else hip.dev_f32(fc.bias, "fc bias")
    if not idx.is_cuda:
        raise hip.XnrsHipError("category indices must live on the HIP device")
    ids = idx.to(torch.int32).contiguous()
    M, K, N = ids.numel(), tab.shape[1], w.shape[0]
    y = torch.empty(tuple(idx.shape) + (N,), dtype=torch.float32, device=tab.device)
    hip.check(hip.lib().xnrs_linear_fwd(hip.ptr(tab), hip.ptr(ids), 1, hip.ptr(w), hip.ptr(b), hip.ptr(y), M, N, K,
                                        hip.ACT_NONE, hip.stream_ptr(tab.device)), "xnrs_linear_fwd(gather)")
    return y
