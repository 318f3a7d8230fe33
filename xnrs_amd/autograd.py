"""Autograd bridge: torch.autograd.Function wrappers around the training entry points of
libxnrs_hip.so (forward that keeps its activations + hand-written backward kernels).

Needed by the grad step (xnrs/training.py:402-431: loss.backward(); Adam) and by integrated
gradients (xnrs/explain.py:160-166: d score / d x).  Everything differentiable on the hot path goes
through HIP kernels; torch only routes gradients between the Functions.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Optional

import torch

from . import hip

_ATT_NAMES = ("q_linear.weight", "q_linear.bias", "k_linear.weight", "k_linear.bias",
              "v_linear.weight", "v_linear.bias", "out.weight", "out.bias")


def _att_tensors(att) -> List[torch.Tensor]:
    return [att.q_linear.weight, att.q_linear.bias, att.k_linear.weight, att.k_linear.bias,
            att.v_linear.weight, att.v_linear.bias, att.out.weight, att.out.bias]


def _pool_tensors(pool) -> List[torch.Tensor]:
    return [pool.fc1.weight, pool.fc1.bias, pool.fc2.weight, pool.fc2.bias]


def _head_tensors(head) -> List[Optional[torch.Tensor]]:
    return [head[0].weight, head[0].bias, head[2].weight, head[2].bias]


class _Cfg:
    """Static (non-tensor) description of one sequence-encoder call."""

    def __init__(self, n_seq, L, D, n_heads, scaled, pool_kind, A, has_head, E, head_bias, dropout_p, seed, want_a):
        self.n_seq, self.L, self.D = n_seq, L, D
        self.n_heads, self.scaled = n_heads, scaled
        self.pool_kind, self.A = pool_kind, A
        self.has_head, self.E, self.head_bias = has_head, E, head_bias
        self.dropout_p, self.seed = dropout_p, seed
        self.want_a = want_a
        self.head_act = hip.ACT_RELU
        from . import ops
        self.seed_word = ops.dropout_seed_word()  # as of the forward: the backward recomputes the same mask


def _param_structs(cfg: _Cfg, params: List[Optional[torch.Tensor]]):
    """Split the flat parameter list (att 8 | pool 4 | head 4) back into the C structs."""
    i = 0
    ap = pp = hp = None
    keep = []
    if cfg.n_heads > 0:
        ts = [hip.dev_f32(t, "mha weight") for t in params[i:i + 8]]
        ap = hip.MhaParams(*[t.data_ptr() for t in ts], cfg.n_heads, 1 if cfg.scaled else 0, float(cfg.dropout_p),
                           int(cfg.seed))
        if cfg.seed_word is not None and cfg.dropout_p > 0:
            ap.seed_dev = cfg.seed_word.data_ptr()  # (ops.set_dropout_seed_word: a fresh draw per hipGraph replay)
            ts = ts + [cfg.seed_word]
        keep += ts
        i += 8
    if cfg.pool_kind == hip.POOL_ADDITIVE:
        ts = [hip.dev_f32(t, "additive weight") for t in params[i:i + 4]]
        pp = hip.AdditiveParams(*[t.data_ptr() for t in ts], cfg.A)
        keep += ts
        i += 4
    if cfg.has_head:
        ts = [None if t is None else hip.dev_f32(t, "head weight") for t in params[i:i + 4]]
        hp = hip.HeadParams(*[None if t is None else t.data_ptr() for t in ts], cfg.E, cfg.head_act)
        keep += ts
        i += 4
    return ap, pp, hp, keep


def _ref(s):
    return None if s is None else C.byref(s)


#: forward and backward over the unmasked token rows only (exact; include/xnrs_hip.h: xnrs_seq_encoder_fwd_train_live /
#: xnrs_seq_encoder_bwd_live).  Environment
#: XNRS_BWD_LIVE_ROWS=0 turns it off; it is used when at most LIVE_ROWS_MAX_FRACTION of the rows are unmasked.
LIVE_ROWS = os.environ.get("XNRS_BWD_LIVE_ROWS", "1") != "0"
LIVE_ROWS_MAX_FRACTION = 0.9
LIVE_ROWS_MIN = 4096  # token rows from which the live-row path pays for its index bookkeeping (tests lower it)
#: how many training forwards took the live-row path (tests assert that the branch they mean to cover really ran)
STATS = {"live_row_forwards": 0, "kv_row_forwards": 0, "shared_qkv_forwards": 0, "device_list_forwards": 0,
         "deferred_dqkv_backwards": 0, "merged_dqkv_backwards": 0, "shared_output_forwards": 0}
#: the row lists built ON THE DEVICE (include/xnrs_hip.h: xnrs_build_row_lists, xnrs_row_lists::counts_dev): no torch
#: bookkeeping and no host read of the counts -- the grad step has no host synchronisation and can be captured in a hipGraph.
#: Taken whenever the shape allows (fp32 GEMM mode, D and A multiples of 4, 16-byte aligned operands); the lists are then
#: used whatever fraction of the rows is live (LIVE_ROWS_MAX_FRACTION needs the count on the host).  XNRS_DEVICE_LISTS=0:
#: the host bookkeeping of round 3 (one .tolist() per encoder call).
DEVICE_LISTS = os.environ.get("XNRS_DEVICE_LISTS", "1") != "0"
#: K|V projection and dWk / dWv over the token rows of the non-empty news only (rides on the live-row path, exact;
#: include/xnrs_hip.h: xnrs_row_lists).  XNRS_KV_ROWS=0 turns it off.
KV_ROWS = os.environ.get("XNRS_KV_ROWS", "1") != "0"


def _addr(t):
    return None if t is None else t.data_ptr()


#: The reference's train step encodes the history twice (training.py:406 scores, :409 get_user_embeddings); with input
#: dropout 0 (every shipped config) the two encodes differ only in their attention-dropout draws, so their Q|K|V images are
#: the same numbers.  A training forward of an attention tower registers its saved blob and row lists here under the
#: identity of everything the image depends on -- data pointer AND version counter of input, mask, ids and the six projection
#: tensors -- and a second forward that finds a LIVE entry reads that image instead of projecting again (and reuses the row
#: lists: no second host read).  Weak references: an entry dies with the first forward's graph.  Bitwise the same step.
#: XNRS_SHARE_QKV=0 turns it off.
SHARE_QKV = os.environ.get("XNRS_SHARE_QKV", "1") != "0"
_QKV_IMAGES = {}
#: ... and their backwards share ONE weight-gradient product per projection: dW = (dQKV_1 + dQKV_2)^T . X (the input X is
#: the same).  Whichever of the two autograd nodes runs first leaves its dQ|dK|dV image in a buffer and returns no gradient
#: for wq/bq/wk/bk/wv/bv (XNRS_DQKV_DEFER) -- but only when the engine says the other node WILL run in this very backward
#: pass (torch._C._will_engine_execute_node) and has not run yet; the second node adds its own dQ|dK|dV to the image and
#: computes the six gradients from the sum (XNRS_DQKV_MERGE).  Same gradients up to summation order.  XNRS_MERGE_DW=0: off.
MERGE_DW = os.environ.get("XNRS_MERGE_DW", "1") != "0"


#: torch.autograd.grad(score, inputs=[tokens]) -- integrated gradients, explain.py:160-166 -- needs no parameter gradient, but
#: ctx.needs_input_grad only says that the parameters REQUIRE grad.  The engine knows what this pass computes: a parameter
#: whose AccumulateGrad node it will not execute gets no weight-gradient product (two thirds of the backward of an IG step).
#: A plain loss.backward() executes every node: nothing changes there.  XNRS_SKIP_UNUSED_DW=0: off.
SKIP_UNUSED_DW = os.environ.get("XNRS_SKIP_UNUSED_DW", "1") != "0"


def _wanted_inputs(ctx, is_tensor, first):
    """[will the engine use the gradient of forward input i in THIS backward pass?] for the inputs from `first` on.
    is_tensor: per forward input, whether a tensor was passed -- ctx.next_functions holds one edge per TENSOR input, in order
    (non-tensor arguments and None have none)."""
    count = len(is_tensor) - first
    if not SKIP_UNUSED_DW or torch._C._current_graph_task_id() == -1:
        return [True] * count
    try:
        nf = ctx.next_functions
        if len(nf) != sum(1 for t in is_tensor if t):
            return [True] * count  # (not the layout this rule assumes: compute everything)
        out = []
        e = 0
        for i, t in enumerate(is_tensor):
            fn = nf[e][0] if t else None
            e += 1 if t else 0
            if i >= first:
                out.append(fn is not None and bool(torch._C._will_engine_execute_node(fn)))
        return out
    except (RuntimeError, AttributeError, IndexError):  # (an engine without the query: compute everything)
        return [True] * count


class _Pair:
    """The two training forwards that share one Q|K|V image, as seen by their backwards."""
    __slots__ = ("a", "b", "image", "task", "owner", "ran", "__weakref__")

    def __init__(self, a, b):
        import weakref
        self.a, self.b = weakref.ref(a), weakref.ref(b)
        self.image = None   # the deferred dQ|dK|dV of the node that ran first
        self.task = -1      # ... and the backward pass (graph task id) it belongs to
        self.owner = 0      # ... and that node's id
        self.ran = {}       # node id -> graph task id of its last backward

    def other(self, ctx):
        a, b = self.a(), self.b()
        return b if ctx is a else a


def _ident(t):
    return None if t is None else (t.data_ptr(), t._version, tuple(t.shape))


def _where(dev):
    """Stream and capture status of a call: a tensor of an EAGER step must never be shared into a step that is being captured
    in a hipGraph (the capture would bake in a pointer to memory outside the graph's pool, freed with the eager step's
    graph), nor across streams."""
    return (torch.cuda.current_stream(dev).cuda_stream, torch.cuda.is_current_stream_capturing())


def _qkv_key(cfg, x, m, ids, params):
    return (_where(x.device), cfg.n_seq, cfg.L, cfg.D, cfg.n_heads, cfg.A, cfg.E, cfg.pool_kind, cfg.has_head, x.device.index, _ident(x), _ident(m),
            _ident(ids), tuple(_ident(p) for p in params[:6]))  # (_att_tensors order: wq, bq, wk, bk, wv, bv, wo, bo)


#: A tower's parameters are shared by every call of the tower (the news tower of a grad step is called for the candidates, the
#: history and the history again: training.py:406,409), and autograd adds the calls' parameter gradients pairwise as they
#: arrive -- one elementwise launch per parameter and extra call, 38 per NRMS step.  Here the calls that are NOT the last of
#: their tower in a backward pass keep their parameter gradients back (they return None for them) and the last one -- the
#: engine says which nodes are still to run: torch._C._will_engine_execute_node -- adds the kept sets to its own with ONE
#: multi-tensor launch each (torch._foreach_add_) and returns the sums.  The same gradients up to the order of the additions;
#: a hook on a parameter sees the sum once instead of each contribution.  XNRS_SUM_SHARED_GRADS=0: off.
SUM_SHARED_GRADS = os.environ.get("XNRS_SUM_SHARED_GRADS", "1") != "0"
_PARAM_GROUPS = {}


class _ParamGroup:
    """The live forward nodes of one tower (= one tuple of parameter tensors) and, during a backward pass, the gradient sets
    its earlier nodes kept back."""
    __slots__ = ("nodes", "kept", "task", "ran")

    def __init__(self):
        self.nodes, self.kept, self.task, self.ran = [], [], -1, {}


def _join_group(ctx, params):
    import weakref
    if not SUM_SHARED_GRADS:
        return None
    key = tuple(None if p is None else id(p) for p in params)
    if len(_PARAM_GROUPS) > 64:
        for k in [k for k, g in _PARAM_GROUPS.items() if not any(r() is not None for r in g.nodes)]:
            del _PARAM_GROUPS[k]
    g = _PARAM_GROUPS.get(key)
    if g is None:
        g = _PARAM_GROUPS[key] = _ParamGroup()
    g.nodes = [r for r in g.nodes if r() is not None]
    g.nodes.append(weakref.ref(ctx))
    return g


def _sum_with_group(ctx, grads):
    """Keep this node's parameter gradients back if another node of its tower is still to run in this pass; otherwise add what
    the earlier nodes kept.  -> the list to return to autograd."""
    g = getattr(ctx, "pgroup", None)
    task = torch._C._current_graph_task_id()
    if g is None or task == -1:
        return grads
    if g.task != task:  # (sets left by a pass that was abandoned are dropped)
        g.kept, g.task = [], task
    g.ran = {k: v for k, v in g.ran.items() if v == task}
    g.ran[id(ctx)] = task
    pending = False
    for r in g.nodes:
        n = r()
        if n is None or n is ctx or g.ran.get(id(n)) == task:
            continue
        try:
            if torch._C._will_engine_execute_node(n):
                pending = True
                break
        except RuntimeError:
            pass
    if pending:
        if any(t is not None for t in grads):
            g.kept.append(grads)
        return [None] * len(grads)
    for kept in g.kept:
        a, b = [], []
        for j, t in enumerate(kept):
            if t is None:
                continue
            if grads[j] is None:
                grads[j] = t          # only an earlier node computed this one (a deferring node leaves its six to the merging one)
            else:
                a.append(grads[j])
                b.append(t)
        if a:
            torch._foreach_add_(a, b)
    g.kept = []
    return grads


#: The folded fc1 pair (W1.Wo, W1.bo + b1: DESIGN.md section 4.6) of a training forward is the same for every call with the
#: same weights: a grad step calls the news tower three times and the user tower twice, so it is computed once per tower and
#: weight version (xnrs_fold_weights; the same routine the in-call fold runs: the same bits) and handed to the forward AND
#: its backward through xnrs_additive_params.w1_folded.  Keyed like every cache here on tensor identity + version counter +
#: stream / capture status.  XNRS_FOLD_TRAIN_CACHE=0: every call folds for itself (three short launches).
FOLD_TRAIN_CACHE = os.environ.get("XNRS_FOLD_TRAIN_CACHE", "1") != "0"
_TRAIN_FOLDS = {}


def _train_fold(l, cfg, params, ap, pp, dev):
    """-> (w1f, b1f) tensors or None when this call does not fold (no attention + additive pair, fold switched off)."""
    if not FOLD_TRAIN_CACHE or cfg.n_heads <= 0 or cfg.pool_kind != hip.POOL_ADDITIVE or not l.xnrs_train_fold_enabled():
        return None
    src = (params[6], params[7], params[8], params[9])  # out.weight, out.bias, fc1.weight, fc1.bias
    key = (_where(dev), dev.index, cfg.D, cfg.A) + tuple(None if t is None else (id(t),) + _ident(t) for t in src)
    hit = _TRAIN_FOLDS.get(key)
    if hit is not None:
        return hit
    if len(_TRAIN_FOLDS) >= 8:  # (every optimizer step retires the entries of the step before)
        _TRAIN_FOLDS.clear()
    w1f = torch.empty((cfg.A, cfg.D), dtype=torch.float32, device=dev)
    b1f = torch.empty((cfg.A,), dtype=torch.float32, device=dev)
    nws = l.xnrs_fold_weights_workspace_bytes(cfg.D, cfg.A)
    ws = hip.workspace(dev, nws)
    hip.check(l.xnrs_fold_weights(C.byref(ap), C.byref(pp), cfg.D, hip.ptr(w1f), hip.ptr(b1f), hip.ptr(ws), nws, hip.stream_ptr(dev)),
              "xnrs_fold_weights")
    _TRAIN_FOLDS[key] = (w1f, b1f)
    return w1f, b1f


class _SeqEncode(torch.autograd.Function):
    """y = head(pool(att(x)))  (any stage optional) with saved activations for the HIP backward."""

    @staticmethod
    def forward(ctx, cfg: _Cfg, x, m, ids, *params):
        x = hip.dev_f32(x, "encoder input")
        dev = x.device
        n, L, D = cfg.n_seq, cfg.L, cfg.D
        ap, pp, hp, keep = _param_structs(cfg, list(params))
        pooled = cfg.pool_kind != hip.POOL_NONE
        Eo = cfg.E if cfg.has_head else D
        y = torch.empty((n, Eo) if pooled else (n, L, D), dtype=torch.float32, device=dev)
        a = torch.empty((n, L), dtype=torch.float32, device=dev) if (cfg.want_a and cfg.pool_kind == hip.POOL_ADDITIVE) else None
        hm = torch.empty((n,), dtype=torch.float32, device=dev) if (pooled and m is not None) else None
        l = hip.lib()
        ctx.fold_pair = _train_fold(l, cfg, params, ap, pp, dev)
        if ctx.fold_pair is not None:
            pp.w1_folded, pp.b1_folded = ctx.fold_pair[0].data_ptr(), ctx.fold_pair[1].data_ptr()
        nsaved = l.xnrs_seq_encoder_saved_bytes(n, L, D, cfg.A, Eo, cfg.n_heads, cfg.pool_kind, int(cfg.has_head))
        saved = torch.empty(max(nsaved, 1), dtype=torch.uint8, device=dev)
        # Nothing that flows through a masked token row reaches the output or a gradient (its pooling weight is
        # exp(e)*0), so the row-parallel products of an attention tower -- forward (query projection, output projection,
        # fc1) and backward -- run over the unmasked rows only (xnrs_seq_encoder_fwd_train_live / _bwd_live).  Index
        # bookkeeping with torch (one host sync for the count); skipped when few rows are masked.
        live = live_src = kv = kv_src = counts = None
        n_live = n_kv = 0
        shared = None  # (blob of an earlier forward over the same input and projection weights, its row lists, its node)
        key = None
        if SHARE_QKV and cfg.n_heads > 0:
            key = _qkv_key(cfg, x, m, ids, params)
            ent = _QKV_IMAGES.get(key)
            if ent is not None:
                blob = ent[0]()
                if blob is None:
                    del _QKV_IMAGES[key]
                else:
                    shared = (blob, ent[1], ent[2]())
        want_lists = LIVE_ROWS and m is not None and cfg.pool_kind == hip.POOL_ADDITIVE and n * L >= LIVE_ROWS_MIN
        if shared is not None:
            live, live_src, n_live, kv, kv_src, n_kv, counts = shared[1]
            STATS["shared_qkv_forwards"] += 1
        elif want_lists and DEVICE_LISTS and _device_lists_ok(cfg, x, params):
            live, live_src, kv, kv_src, counts = _device_lists(l, cfg, m, ids, dev)
            n_live = n_kv = n * L  # capacities: the counts stay on the device
            STATS["device_list_forwards"] += 1
            STATS["live_row_forwards"] += 1
            if KV_ROWS and cfg.n_heads > 0:
                STATS["kv_row_forwards"] += 1
            else:
                kv = kv_src = None
        elif want_lists:
            live, live_src, n_live, kv, kv_src, n_kv = _host_lists(cfg, m, ids, dev)
        qkv_shared = None
        if shared is not None:
            qkv_shared = shared[0].data_ptr() + l.xnrs_seq_encoder_saved_qkv_offset(n, L, D, cfg.A, Eo, cfg.n_heads, cfg.pool_kind,
                                                                                     int(cfg.has_head))
        lists = None
        if live is not None or qkv_shared is not None:
            lists = hip.RowLists(_addr(live), _addr(live_src), n_live, _addr(kv), _addr(kv_src), n_kv, qkv_shared,
                                 _addr(counts), None, hip.DQKV_OWN)
        hip.check(l.xnrs_seq_encoder_fwd_train_rows(hip.ptr(x), hip.ptr(m), hip.ptr(ids), n, L, D, _ref(ap), cfg.pool_kind,
                                                    _ref(pp), _ref(hp), hip.ptr(y), hip.ptr(a), hip.ptr(hm), hip.ptr(saved),
                                                    nsaved, _ref(lists), hip.stream_ptr(dev)),
                  "xnrs_seq_encoder_fwd_train_rows")
        ctx.row_lists = (live, live_src, n_live, kv, kv_src, n_kv, counts)
        ctx.qkv_shared = qkv_shared
        ctx.qkv_owner = shared[0] if shared is not None else None  # keeps the other forward's blob alive until our backward
        ctx.pair = None
        if shared is not None and shared[2] is not None and getattr(shared[2], "pair", None) is None:
            ctx.pair = shared[2].pair = _Pair(shared[2], ctx)  # the two backwards share one dW product per projection
        if key is not None and shared is None:
            import weakref
            for k in [k for k, v in _QKV_IMAGES.items() if v[0]() is None]:
                del _QKV_IMAGES[k]
            _QKV_IMAGES[key] = (weakref.ref(saved), ctx.row_lists, weakref.ref(ctx))
        ctx.pgroup = _join_group(ctx, params)
        ctx.fold = l.xnrs_train_fold_enabled()  # the saved blob is laid out by this decision (include/xnrs_hip.h)
        ctx.cfg = cfg
        ctx.nsaved = nsaved
        ctx.n_params = len(params)
        ctx.save_for_backward(x, m, ids, saved, *[p for p in params if p is not None])
        ctx.param_none = [p is None for p in params]
        outs = [y, a if a is not None else y.new_empty(0), hm if hm is not None else y.new_empty(0)]
        ctx.mark_non_differentiable(outs[1], outs[2])
        ctx.set_materialize_grads(False)  # (the two side outputs never carry a gradient: no zero tensors are made for them)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dy, _da, _dhm):
        _OUTPUTS.pop(getattr(ctx, "okey", None), None)  # the graph is being consumed: its outputs are no longer shareable
        if dy is None:  # (nothing flows into the only differentiable output)
            return (None,) * (4 + ctx.n_params)
        cfg = ctx.cfg
        x, m, ids, saved, *ptensors = ctx.saved_tensors
        it = iter(ptensors)
        params = [None if isnone else next(it) for isnone in ctx.param_none]
        dev = x.device
        n, L, D = cfg.n_seq, cfg.L, cfg.D
        Eo = cfg.E if cfg.has_head else D
        ap, pp, hp, keep = _param_structs(cfg, params)
        if getattr(ctx, "fold_pair", None) is not None:  # the pair the forward was given (the saved blob holds no copy then)
            pp.w1_folded, pp.b1_folded = ctx.fold_pair[0].data_ptr(), ctx.fold_pair[1].data_ptr()
        dy = hip.dev_f32(dy, "grad output")
        need = ctx.needs_input_grad  # (cfg, x, m, ids, *params)
        want_dx = need[1]
        if want_dx and ids is not None:
            raise hip.XnrsHipError("no input gradient through an id-gathered news table")
        dx = torch.empty((n, L, D), dtype=torch.float32, device=dev) if want_dx else None
        wanted = _wanted_inputs(ctx, [False, True, m is not None, ids is not None] + [p is not None for p in params], 4)
        grads: List[Optional[torch.Tensor]] = []
        for j, p in enumerate(params):
            grads.append(torch.empty_like(p, dtype=torch.float32) if (p is not None and need[4 + j] and wanted[j]) else None)
        i = 0
        ga = gp = gh = None
        if cfg.n_heads > 0:
            ga = hip.MhaGrads(*[None if g is None else g.data_ptr() for g in grads[i:i + 8]])
            i += 8
        if cfg.pool_kind == hip.POOL_ADDITIVE:
            gp = hip.AdditiveGrads(*[None if g is None else g.data_ptr() for g in grads[i:i + 4]])
            i += 4
        if cfg.has_head:
            gh = hip.HeadGrads(*[None if g is None else g.data_ptr() for g in grads[i:i + 4]])
            i += 4
        l = hip.lib()
        if l.xnrs_train_fold_enabled() != ctx.fold:
            raise hip.XnrsHipError("XNRS_FOLD_TRAIN changed between a training forward and its backward: the saved "
                                   "activations were laid out for the other setting (reload the knobs outside a step)")
        nws = l.xnrs_seq_encoder_bwd_workspace_bytes(n, L, D, cfg.A, Eo, cfg.n_heads, cfg.pool_kind, int(cfg.has_head))
        ws = hip.workspace(dev, nws)
        live, live_src, n_live, kv, kv_src, n_kv, counts = ctx.row_lists  # the row lists built by the forward
        # one dW product per projection for the two backwards over one Q|K|V image (see MERGE_DW above)
        mode, image = hip.DQKV_OWN, None
        pair = ctx.pair
        if pair is not None:
            task = torch._C._current_graph_task_id()
            qkv_all = MERGE_DW and not want_dx and all(g is not None for g in grads[:6])
            if pair.image is not None and pair.task == task and pair.owner != id(ctx) and qkv_all:
                mode, image, pair.image = hip.DQKV_MERGE, pair.image, None
                STATS["merged_dqkv_backwards"] += 1
            else:
                pair.image = None  # (an image left by a pass whose second node never ran is dropped)
                partner = pair.other(ctx)
                if (qkv_all and partner is not None and task != -1 and pair.ran.get(id(partner)) != task
                        and torch._C._will_engine_execute_node(partner)):
                    mode, image = hip.DQKV_DEFER, torch.empty((n * L, 3 * D), dtype=torch.float32, device=dev)
                    pair.image, pair.task, pair.owner = image, task, id(ctx)
                    for j in range(6):
                        grads[j] = None  # the merging node returns them (computed from the sum)
                    STATS["deferred_dqkv_backwards"] += 1
            pair.ran[id(ctx)] = task
            if cfg.n_heads > 0:
                ga = hip.MhaGrads(*[None if g is None else g.data_ptr() for g in grads[0:8]])
        lists = None
        if live is not None or ctx.qkv_shared is not None or mode != hip.DQKV_OWN:
            lists = hip.RowLists(_addr(live), _addr(live_src), n_live, _addr(kv), _addr(kv_src), n_kv, ctx.qkv_shared,
                                 _addr(counts), _addr(image), mode)
        hip.check(l.xnrs_seq_encoder_bwd_rows(hip.ptr(x), hip.ptr(m), hip.ptr(ids), n, L, D, _ref(ap), cfg.pool_kind, _ref(pp),
                                              _ref(hp), hip.ptr(saved), ctx.nsaved, hip.ptr(dy), hip.ptr(dx), _ref(ga),
                                              _ref(gp), _ref(gh), _ref(lists), hip.ptr(ws), nws,
                                              hip.stream_ptr(dev)), "xnrs_seq_encoder_bwd_rows")
        return (None, dx, None, None, *_sum_with_group(ctx, grads))


def _host_lists(cfg, m, ids, dev):
    """The row lists by torch bookkeeping: a nonzero and ONE host read for both counts (round 3's path; kept for shapes the
    device-counted kernels do not serve and behind XNRS_DEVICE_LISTS=0)."""
    n, L = cfg.n_seq, cfg.L
    live = live_src = kv = kv_src = None
    n_kv = 0
    lm = (m[ids.long()] if ids is not None else m).reshape(n, L).ne(0)
    news_live = lm.any(dim=1)
    n_live, n_news_live = (int(v) for v in torch.stack([lm.sum(), news_live.sum()]).tolist())
    if n_live > LIVE_ROWS_MAX_FRACTION * n * L:
        return None, None, 0, None, None, 0
    rows_live = torch.nonzero_static(lm.reshape(-1), size=n_live).squeeze(1)
    live = rows_live.to(torch.int32)
    STATS["live_row_forwards"] += 1
    src_news = ids.long() if ids is not None else None
    if ids is not None:
        seq = torch.div(rows_live, L, rounding_mode="floor")
        live_src = (src_news[seq] * L + (rows_live - seq * L)).to(torch.int32)
    # the token rows of the non-empty news: K and V are projected (and their weight gradients summed) over
    # these only -- nobody reads the keys of a news without a live query (include/xnrs_hip.h: xnrs_row_lists)
    if KV_ROWS and cfg.n_heads > 0 and n_news_live < n:
        news_idx = torch.nonzero_static(news_live, size=n_news_live).squeeze(1)
        tok = torch.arange(L, device=dev)
        kv = (news_idx[:, None] * L + tok[None, :]).reshape(-1).to(torch.int32)
        n_kv = n_news_live * L
        STATS["kv_row_forwards"] += 1
        if ids is not None:
            kv_src = (src_news[news_idx][:, None] * L + tok[None, :]).reshape(-1).to(torch.int32)
    return live, live_src, n_live, kv, kv_src, n_kv


def _device_lists_ok(cfg, x, params) -> bool:
    """Do the kernels that read their row count on the device serve this call?  (The conditions api.hip checks again:
    fp32 GEMM mode, D and A multiples of 4, 16-byte aligned input and weights.)"""
    if hip.get_gemm_mode() != 0 or cfg.D % 4 != 0 or cfg.A % 4 != 0 or cfg.n_seq * cfg.L >= 2 ** 31:
        return False
    if x.data_ptr() % 16 != 0:
        return False
    return all(p is None or p.data_ptr() % 16 == 0 for p in params)


def _device_lists(l, cfg, m, ids, dev):
    n, L = cfg.n_seq, cfg.L
    cap = n * L
    buf = torch.empty((4 if ids is not None else 2, cap), dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    live, kv = buf[0], buf[1]
    live_src, kv_src = (buf[2], buf[3]) if ids is not None else (None, None)
    nws = l.xnrs_row_lists_workspace_bytes(n)
    ws = hip.workspace(dev, nws)
    hip.check(l.xnrs_build_row_lists(hip.ptr(m), hip.ptr(ids), n, L, hip.ptr(live), hip.ptr(live_src), hip.ptr(kv),
                                     hip.ptr(kv_src), hip.ptr(counts), hip.ptr(ws), nws, hip.stream_ptr(dev)),
              "xnrs_build_row_lists")
    return live, live_src, kv, kv_src, counts


#: A DETERMINISTIC encode (no attention dropout in play: no attention tower, or p = 0 / eval mode) called again with the very
#: same tensors -- input, mask, ids and every parameter the same objects at the same version -- while the first call's graph
#: is still alive returns the first call's OUTPUT TENSORS (same autograd node).  The reference's train step encodes the
#: history twice (training.py:406 model(batch), :409 get_user_embeddings(batch)); for StandardRec / NAML (no dropout anywhere
#: in the shipped configs) the second encode is bit for bit the first, so it is not computed, and since the user tower then
#: sees the same tensor object again, neither is its second call.  Gradients: the node receives the sum of both uses'
#: gradients and runs ONE backward -- the same gradient up to summation order.  XNRS_SHARE_OUTPUTS=0: off.
SHARE_OUTPUTS = os.environ.get("XNRS_SHARE_OUTPUTS", "1") != "0"
_OUTPUTS = {}
_OUTPUTS_HORIZON = 16  # encoder calls after which an unclaimed entry is dropped (a step's second encode follows within ~8)
#: ... and a cap on the saved activations the unclaimed entries may pin (an entry keeps its graph, i.e. its saved blob, alive
#: until its backward runs or it is dropped): a forward-only loop that forgot torch.no_grad() must not hoard a blob per call
_OUTPUTS_MAX_BYTES = int(os.environ.get("XNRS_SHARE_OUTPUTS_MB", "4096")) << 20
_TICK = 0


def _root(t):
    """The tensor that owns t's storage (views compare by their base: modules reshape their inputs on every call)."""
    return t if (t is None or t._base is None) else t._base


def _alive(refs, objs):
    return all((r is None and o is None) or (r is not None and r() is _root(o)) for r, o in zip(refs, objs))


def _run(x, m, ids, att, pooler, head, pool_kind, dropout_p, seed, want_a):
    from . import ops
    import weakref
    x_in, m_in, ids_in = x, m, ids
    x = hip.dev_f32(x, "encoder input")
    n_tab, L, D = x.shape
    m2 = ops._mask2d(m, n_tab, L, "encoder mask")
    if ids is not None:
        ids = ids.to(torch.int32).contiguous()
        n = ids.numel()
    else:
        n = n_tab
    params: List[Optional[torch.Tensor]] = []
    n_heads, scaled, A, E, has_head, head_bias = 0, True, 0, D, False, False
    if att is not None:
        params += _att_tensors(att)
        n_heads, scaled = att.h, att.scaled
    if pool_kind == hip.POOL_ADDITIVE:
        params += _pool_tensors(pooler)
        A = pooler.fc1.out_features
    if head is not None:
        params += _head_tensors(head)
        has_head, E = True, head[0].out_features
    cfg = _Cfg(n, L, D, n_heads, scaled, pool_kind, A, has_head, E, head_bias, dropout_p, seed, want_a)
    if head is not None:
        cfg.head_act = hip.head_activation(head[1])
    okey = None
    global _TICK
    _TICK += 1
    for k in [k for k, v in _OUTPUTS.items() if _TICK - v[0] > _OUTPUTS_HORIZON]:
        del _OUTPUTS[k]  # (never claimed: a forward in grad mode that nobody repeated)
    if SHARE_OUTPUTS and (n_heads == 0 or dropout_p == 0.0) and torch.is_grad_enabled():
        objs = [x_in, m_in, ids_in] + params
        okey = (_where(x.device), n, L, D, n_heads, scaled, pool_kind, A, has_head, E, cfg.head_act, want_a, x.device.index, _ident(x), _ident(m2),
                _ident(ids), tuple(_ident(p) for p in params))
        hit = _OUTPUTS.get(okey)
        if hit is not None:
            _, refs, (y, a, hm), ver, _nb = hit
            if _alive(refs, objs) and y._version == ver and y.grad_fn is not None:
                STATS["shared_output_forwards"] += 1
                hit[0] = _TICK
                return y, a, hm
            del _OUTPUTS[okey]
    y, a, hm = _SeqEncode.apply(cfg, x, m2, ids, *params)
    a = a if a.numel() else None
    hm = hm if hm.numel() else None
    if okey is not None and y.grad_fn is not None:
        # the entry holds the outputs themselves (modules drop them: a view of y fed to torch.cat keeps nothing alive) and
        # is dropped by the node's backward, or after _OUTPUTS_HORIZON further encoder calls
        wr = lambda t: None if t is None else weakref.ref(t)  # noqa: E731
        _OUTPUTS[okey] = [_TICK, [wr(_root(o)) for o in objs], (y, a, hm), y._version, int(getattr(y.grad_fn, "nsaved", 0))]
        y.grad_fn.okey = okey
        pinned = sum(v[4] for v in _OUTPUTS.values())
        for k in sorted(_OUTPUTS, key=lambda k: _OUTPUTS[k][0]):  # oldest first; the entry just made always stays
            if pinned <= _OUTPUTS_MAX_BYTES or k == okey:
                break
            pinned -= _OUTPUTS[k][4]
            del _OUTPUTS[k]
    return y, a, hm


def _pool_kind(pooler):
    from .models.components import layers
    if isinstance(pooler, layers.AdditiveAttention):
        return hip.POOL_ADDITIVE
    if isinstance(pooler, layers.MaskedMean):
        return hip.POOL_MEAN
    raise hip.XnrsHipError(f"pooler {type(pooler).__name__} has no HIP implementation")


# ------------------------------------------------------------------------- entry points used by ops.py
def mha(x, m, att, dropout_p, seed):
    y, _, _ = _run(x, m, None, att, None, None, hip.POOL_NONE, dropout_p, seed, False)
    return y


def additive(x, m, pool, return_weights):
    y, a, _ = _run(x, m, None, None, pool, None, hip.POOL_ADDITIVE, 0.0, 0, return_weights)
    y = y.unsqueeze(1)
    return (y, a.unsqueeze(-1)) if return_weights else y


def masked_mean(x, m):
    from .models.components import layers
    y, _, _ = _run(x, m, None, None, layers.MaskedMean(), None, hip.POOL_MEAN, 0.0, 0, False)
    return y.unsqueeze(1)


def text_encoder(x, m, enc, ids, dropout_p, seed):
    head = getattr(enc, "head", None)
    y, _, hm = _run(x, m, ids, enc.att, enc.pooler, head, _pool_kind(enc.pooler), dropout_p, seed, False)
    return y, hm


def user_encoder(x, m, enc, return_weights, dropout_p, seed):
    head = getattr(enc, "head", None)
    y, a, _ = _run(x, m, None, enc.att, enc.pooler, head, _pool_kind(enc.pooler), dropout_p, seed, return_weights)
    y = y.unsqueeze(1)
    return (y, a.unsqueeze(-1)) if return_weights else y


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        from . import ops
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        with torch.no_grad():
            return ops.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        x = hip.dev_f32(x, "x")
        w = hip.dev_f32(w, "w")
        dy = hip.dev_f32(dy, "dy")
        K, N = w.shape[1], w.shape[0]
        M = x.numel() // K
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = torch.empty(N, dtype=torch.float32, device=x.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        l = hip.lib()
        nws = l.xnrs_linear_bwd_workspace_bytes(M, N, K)
        ws = hip.workspace(x.device, nws)
        hip.check(l.xnrs_linear_bwd(hip.ptr(x), None, 0, hip.ptr(w), hip.ptr(dy), hip.ptr(dx), hip.ptr(dw), hip.ptr(db), M, N,
                                    K, hip.ptr(ws), nws, hip.stream_ptr(x.device)), "xnrs_linear_bwd")
        return dx, dw, db


def linear(x, w, b):
    return _Linear.apply(x, w, b)


class _EmbeddingLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table, w, b):
        tab = hip.dev_f32(table, "embedding table")
        wd = hip.dev_f32(w, "fc weight")
        bd = None if b is None else hip.dev_f32(b, "fc bias")
        idx = ids.to(torch.int32).contiguous()
        M, K, N = idx.numel(), tab.shape[1], wd.shape[0]
        y = torch.empty(tuple(ids.shape) + (N,), dtype=torch.float32, device=tab.device)
        hip.check(hip.lib().xnrs_linear_fwd(hip.ptr(tab), hip.ptr(idx), 1, hip.ptr(wd), hip.ptr(bd), hip.ptr(y), M, N, K,
                                            hip.ACT_NONE, hip.stream_ptr(tab.device)), "xnrs_linear_fwd(gather)")
        ctx.save_for_backward(idx, tab, wd)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, tab, w = ctx.saved_tensors
        dy = hip.dev_f32(dy, "dy")
        M, K, N = idx.numel(), tab.shape[1], w.shape[0]
        need = ctx.needs_input_grad
        d_tab = torch.empty_like(tab) if need[1] else None
        dw = torch.empty_like(w) if need[2] else None
        db = torch.empty(N, dtype=torch.float32, device=tab.device) if (ctx.has_bias and need[3]) else None
        l = hip.lib()
        nws = l.xnrs_embedding_linear_bwd_workspace_bytes(M, N, K)
        ws = hip.workspace(tab.device, nws)
        hip.check(l.xnrs_embedding_linear_bwd(hip.ptr(tab), hip.ptr(idx), hip.ptr(w), hip.ptr(dy), hip.ptr(d_tab), hip.ptr(dw),
                                              hip.ptr(db), M, N, K, tab.shape[0], hip.ptr(ws), nws,
                                              hip.stream_ptr(tab.device)), "xnrs_embedding_linear_bwd")
        return None, d_tab, dw, db


def embedding_linear(idx, embedder, fc):
    if not idx.is_cuda:
        raise hip.XnrsHipError("category indices must live on the HIP device")
    return _EmbeddingLinear.apply(idx, embedder.weight, fc.weight, fc.bias)


class _DotScoring(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, c, normalize):
        from . import ops
        u = hip.dev_f32(u, "user vector")
        c = hip.dev_f32(c, "candidate vectors")
        ctx.save_for_backward(u, c)
        ctx.normalize = bool(normalize)
        with torch.no_grad():
            return ops.dot_scoring_forward(u, c, ctx.normalize)

    @staticmethod
    def backward(ctx, dr):
        u, c = ctx.saved_tensors
        dr = hip.dev_f32(dr, "dr")
        B, Cn, E = c.shape
        du = torch.empty_like(u) if ctx.needs_input_grad[0] else None
        dc = torch.empty_like(c) if ctx.needs_input_grad[1] else None
        fn = hip.lib().xnrs_dot_scoring_norm_bwd if ctx.normalize else hip.lib().xnrs_dot_scoring_bwd
        hip.check(fn(hip.ptr(u), hip.ptr(c), hip.ptr(dr), hip.ptr(du), hip.ptr(dc), B, Cn, E, hip.stream_ptr(c.device)),
                  "xnrs_dot_scoring_norm_bwd" if ctx.normalize else "xnrs_dot_scoring_bwd")
        return du, dc, None


def dot_scoring(u, c, normalize):
    return _DotScoring.apply(u, c, normalize)
