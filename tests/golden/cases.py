"""Case table shared by the golden generator (make_golden.py) and the tests.

Pure data + tiny helpers; no reference import.  A=256 is the additive-attention hidden size
hard-coded by the reference assemblies (xnrs/models/full_models/nrms.py:18,34).
"""
import numpy as np
import torch


def block_mask(rng, B, N):
    """(B,N,1) 0/1 mask: row 0 partially masked, last row of the batch fully masked when B>1."""
    L = rng.integers(1, N + 1, size=(B,))
    m = (np.arange(N)[None, :] < L[:, None]).astype(np.float32)
    if B > 1:
        m[-1] = 0.0
    return m[..., None].copy()


BLOCKS = {
    # AdditiveAttention (xnrs/models/components/layers.py:40-69)
    "add_tiny": dict(kind="additive", B=2, N=8, D=32, A=256, mask=True, seed=100),
    "add_nomask": dict(kind="additive", B=3, N=4, D=16, A=256, mask=False, seed=101),
    "add_768": dict(kind="additive", B=2, N=50, D=768, A=256, mask=True, seed=102),
    "add_odd": dict(kind="additive", B=3, N=7, D=20, A=24, mask=True, seed=103),
    # MultiHeadAttention (layers.py:105-156)
    "mha_tiny": dict(kind="mha", B=2, N=8, D=32, h=4, mask=True, seed=110),
    "mha_nomask": dict(kind="mha", B=2, N=5, D=32, h=4, mask=False, seed=111),
    "mha_300_15": dict(kind="mha", B=3, N=30, D=300, h=15, mask=True, seed=112),
    "mha_320_16": dict(kind="mha", B=2, N=30, D=320, h=16, mask=True, seed=113),
    "mha_768_16": dict(kind="mha", B=2, N=50, D=768, h=16, mask=True, seed=114),
    "mha_user": dict(kind="mha", B=4, N=25, D=256, h=16, mask=True, seed=115),
    "mha_dk6": dict(kind="mha", B=2, N=9, D=18, h=3, mask=True, seed=116),
    "mha_s70": dict(kind="mha", B=2, N=70, D=64, h=4, mask=True, seed=117),
    # MaskedMean (layers.py:19-37)
    "mean_tiny": dict(kind="mean", B=3, N=8, D=32, seed=120),
    # DotScoring (scoring.py:6-23)
    "dot_tiny": dict(kind="dot", B=3, N=5, D=16, normalize=False, seed=130),
    "dot_norm": dict(kind="dot", B=3, N=5, D=256, normalize=True, seed=131),
}

ENCODERS = {
    # TextEncoder (news_encoding.py:8-60): att / pooler / head / bias variants
    "news_nrms_tiny": dict(tower="news", B=2, N=3, S=8, D=32, h=4, A=256, E=16, att=True, pooler="additive", head=True, bias=True, seed=200),
    "news_add_head_nobias": dict(tower="news", B=2, N=3, S=8, D=32, h=4, A=256, E=16, att=False, pooler="additive", head=True, bias=False, seed=201),
    "news_add_nohead": dict(tower="news", B=2, N=3, S=8, D=32, h=4, A=256, E=32, att=False, pooler="additive", head=False, bias=True, seed=202),
    "news_mean_head": dict(tower="news", B=2, N=3, S=8, D=32, h=4, A=256, E=16, att=False, pooler="mean", head=True, bias=True, seed=203),
    "news_nrms_300": dict(tower="news", B=2, N=4, S=30, D=300, h=15, A=256, E=256, att=True, pooler="additive", head=True, bias=True, seed=204),
    "news_nrms_768": dict(tower="news", B=1, N=3, S=50, D=768, h=16, A=256, E=256, att=True, pooler="additive", head=True, bias=True, seed=205),
    # UserEncoder (user_encoding.py:6-81)
    "user_nrms": dict(tower="user", B=3, N=25, D=256, h=16, A=256, E=256, att=True, pooler="additive", head=False, bias=True, seed=210),
    "user_std_head": dict(tower="user", B=3, N=7, D=32, h=4, A=256, E=32, att=False, pooler="additive", head=True, bias=False, seed=211),
    "user_mean": dict(tower="user", B=3, N=7, D=32, h=4, A=256, E=32, att=False, pooler="mean", head=False, bias=True, seed=212),
    # shipped shapes (mind_small_*.yml: hist_len 25; BASELINE configs[2]: history 50): outputs only, inputs regenerate
    "user_nrms_h50": dict(tower="user", B=2, N=50, D=256, h=16, A=256, E=256, att=True, pooler="additive", head=False, bias=True, seed=213),
    "user_std_h25": dict(tower="user", B=2, N=25, D=256, h=16, A=256, E=256, att=False, pooler="additive", head=True, bias=False, seed=214),
    "news_add_768": dict(tower="news", B=1, N=3, S=50, D=768, h=16, A=256, E=256, att=False, pooler="additive", head=True, bias=False, seed=206),
}

MODELS = {
    "nrms_tiny": dict(model="NRMS", B=3, H=4, C=3, S=8, D=32, h=4, E=16, bias=False, seed=300),
    "nrms_300": dict(model="NRMS", B=2, H=6, C=5, S=30, D=300, h=15, E=240, bias=False, seed=301, min_len=5),
    "nrms_shipped": dict(model="NRMS", B=1, H=2, C=2, S=50, D=768, h=16, E=256, bias=False, seed=302, min_len=5),
    "standard_tiny": dict(model="standard", B=3, H=4, C=3, S=8, D=32, h=4, E=16, bias=False, seed=310),
    "standard_bias": dict(model="standard", B=2, H=5, C=5, S=12, D=64, h=4, E=32, bias=True, seed=311),
    "base_tiny": dict(model="base", B=3, H=4, C=3, S=8, D=32, h=4, E=16, bias=False, seed=320),
    "naml_tiny": dict(model="NAML", B=2, H=4, C=3, S=8, D=32, h=4, E=16, bias=False, seed=330),
    # the other models at the shipped token shape S=50, D=768, E=256 (config/mind_small_{CL,NAML}.yml; BASELINE configs[3],
    # [4]) and NRMS at BASELINE configs[2]'s history of 50: outputs only (a few hundred floats each)
    "standard_shipped": dict(model="standard", B=1, H=3, C=2, S=50, D=768, h=16, E=256, bias=False, seed=312, min_len=5),
    "base_shipped": dict(model="base", B=1, H=3, C=2, S=50, D=768, h=16, E=256, bias=False, seed=321, min_len=5),
    "naml_shipped": dict(model="NAML", B=1, H=3, C=2, S=50, D=768, h=16, E=256, bias=False, seed=331, min_len=5),
    "nrms_h50": dict(model="NRMS", B=1, H=50, C=5, S=50, D=768, h=16, E=256, bias=False, seed=303, min_len=5),
}

LSTUR = dict(model="LSTUR", B=2, H=4, C=3, S=8, D=32, h=4, E=16, bias=False, seed=340)

GRAD = dict(model="NRMS", B=4, H=3, C=3, S=8, D=32, h=4, E=16, bias=False, seed=400,
            temperature=0.08, lambda_cl=0.1)


# The same train step at the SHIPPED shape (config/mind_small_NRMS.yml: S=50, D=768, 16 heads -> d_k = 48, E=256): the
# d_k = 48 / S = 50 backward kernels (mha_bwd_fused_kernel<3>, the live-row dW path) against the REAL reference.  The full
# gradients are 12.6 MB, so tensors of more than GRAD_SAMPLE_MIN elements are stored as a fixed 4 096-element sample:
# flat indices grad_sample_idx(numel) -- a multiplicative hash walk, the same on every machine.
GRAD_SHIPPED = dict(model="NRMS", B=3, H=3, C=2, S=50, D=768, h=16, E=256, bias=False, seed=410, min_len=5,
                    temperature=0.08, lambda_cl=0.1, themes=["theme1", "theme1", "theme0"])
# the same step for the attention-free bi-encoder of BASELINE configs[3] (config/mind_small_CL.yml: StandardRec, additive
# towers + heads, biases on) at the shipped token shape: pins the live-row grad step of attention-free towers (round 3)
GRAD_SHIPPED_STD = dict(model="standard", B=3, H=4, C=2, S=50, D=768, h=16, E=256, bias=True, seed=430, min_len=5,
                        temperature=0.08, lambda_cl=0.1, themes=["theme2", "theme0", "theme2"])
# ... and for BASELINE configs[4]'s model (NAML: title + abstract views, category / subcategory embeddings, feature pooler)
GRAD_SHIPPED_NAML = dict(model="NAML", B=3, H=3, C=2, S=50, D=768, h=16, E=256, bias=False, seed=450, min_len=5,
                         temperature=0.08, lambda_cl=0.1, themes=["theme1", "theme3", "theme1"])
GRAD_SAMPLE_MIN, GRAD_SAMPLE_N = 8192, 4096


def grad_sample_idx(numel: int):
    """None: the tensor is stored whole; else int64 flat indices of the stored sample."""
    if numel <= GRAD_SAMPLE_MIN:
        return None
    return (np.arange(GRAD_SAMPLE_N, dtype=np.int64) * 2654435761 + 12345) % numel


def grad_sample(t):
    """The stored view of a gradient tensor (numpy or torch): whole, or its fixed sample (flat)."""
    a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    idx = grad_sample_idx(a.size)
    return a if idx is None else a.reshape(-1)[idx]


def model_cfg(c: dict) -> dict:
    """The flat YAML keys make_model reads, at the case's shape (one definition: xnrs_amd.synth.model_cfg)."""
    from xnrs_amd import synth
    return synth.model_cfg(c)


def theme_labels(themes):
    """Deterministic stand-in for training.py:414-417 (string themes -> int labels); the label
    *values* do not matter to the loss, only equality does."""
    uniq = sorted(set(themes))
    idx = {t: i for i, t in enumerate(uniq)}
    return torch.tensor([idx[t] for t in themes])


# ----------------------------------------------------------------------------------------------
# input regeneration (shared by make_golden.py and the tests so both sides see identical bits)
def block_inputs(c):
    """-> (x:(B,N,D), m:(B,N,1), u:(B,1,D) or None) for a BLOCKS case."""
    from xnrs_amd import synth
    rng = synth.rng_for(c["seed"])
    B, N, D = c["B"], c["N"], c["D"]
    x = torch.from_numpy(rng.standard_normal((B, N, D)).astype(np.float32))
    m = torch.from_numpy(block_mask(rng, B, N))
    u = None
    if c["kind"] == "dot":
        u = torch.from_numpy(rng.standard_normal((B, 1, D)).astype(np.float32))
    return x, m, u


def encoder_inputs(c):
    """-> (x, m) for an ENCODERS case: news tower (B,N,S,D)/(B,N,S,1); user tower (B,N,D)/(B,N,1)."""
    from xnrs_amd import synth
    rng = synth.rng_for(c["seed"])
    if c["tower"] == "news":
        return synth.token_block(rng, c["B"], c["N"], c["S"], c["D"], min_len=1, full_pad_prob=0.25)
    x = torch.from_numpy(rng.standard_normal((c["B"], c["N"], c["D"])).astype(np.float32))
    m = torch.from_numpy(block_mask(rng, c["B"], c["N"]))
    return x, m


def model_batch(c):
    from xnrs_amd import synth
    naml = c["model"] == "NAML"
    return synth.make_batch(c["seed"], c["B"], c["H"], c["C"], c["S"], c["D"], min_len=c.get("min_len", 1),
                            abstract=naml, n_categories=19 if naml else 0, n_subcategories=300 if naml else 0)


def lstur_inputs(c=None):
    from xnrs_amd import synth
    c = c or LSTUR
    rng = synth.rng_for(c["seed"])
    x, m = synth.token_block(rng, c["B"], c["H"], c["S"], c["D"])
    ci = torch.from_numpy(rng.integers(0, 19 + 1, size=(c["B"], c["H"])).astype(np.int32))
    si = torch.from_numpy(rng.integers(0, 300 + 1, size=(c["B"], c["H"])).astype(np.int32))
    return x, m, ci, si


def infonce_inputs():
    from xnrs_amd import synth
    rng = synth.rng_for(GRAD["seed"] + 7)
    e = torch.from_numpy(rng.standard_normal((8, 16)).astype(np.float32))
    lab = torch.from_numpy(rng.integers(0, 3, size=(8,)))
    return e, lab


# ----------------------------------------------------------------------------------------------
# data-side cases (device batch assembly / evaluation): a tiny synthetic corpus in the reference's
# in-memory formats (mind.py:161-164 news dict; dataset.py:50-51 sessions)
DATA = dict(n_news=20, S=4, D=8, l_hist=5, n_neg=4, seed=600)


def data_corpus(c=None):
    from xnrs_amd import synth
    c = c or DATA
    rng = synth.rng_for(c["seed"])
    news_feat = {}
    for i in range(c["n_news"]):
        L = int(rng.integers(1, c["S"] + 1))
        emb = rng.standard_normal((1, c["S"], c["D"])).astype(np.float32)
        mask = (np.arange(c["S"])[None, :] < L).astype(np.float32)
        news_feat[f"N{i}"] = {"title_emb": (emb, mask), "category_index": int(rng.integers(1, 6))}
    ids = list(news_feat)
    sessions = []
    for s_i, (nh, npos, nneg) in enumerate([(2, 1, 3), (5, 2, 4), (9, 1, 9), (1, 3, 2), (7, 1, 12), (4, 2, 6)]):
        pick = lambda k: [ids[int(j)] for j in rng.integers(0, len(ids), size=k)]  # noqa: E731
        sessions.append({"history": pick(nh), "positives": pick(npos), "negatives": pick(nneg),
                         "main_theme": f"theme{s_i % 3}", "main_category": "news", "user_index": s_i})
    return news_feat, sessions


# NAML / LSTUR-news by table row (BASELINE configs[4]): a corpus with title + abstract tokens and both category columns in
# the reference's in-memory format; the REAL NewsRecDataset materialises the eval items (dataset.py:48-163) and the REAL
# NAML / LSTURNewsEncoder score them (make_golden.py: naml_data_cases) -- the id path over a NewsStore must reproduce that.
NAML_DATA = dict(n_news=24, S=6, D=32, E=16, l_hist=5, seed=620, model_seed=621)
NAML_SESSIONS = [(2, 1, 3), (5, 2, 4), (9, 1, 6), (1, 2, 2), (7, 1, 8)]


def naml_corpus(c=None):
    from xnrs_amd import synth
    c = c or NAML_DATA
    rng = synth.rng_for(c["seed"])
    news_feat = {}
    for i in range(c["n_news"]):
        feats = {}
        for name, S in (("title_emb", c["S"]), ("abstract_emb", c["S"])):  # one l_seq for every text feature (dataset.py:80-84)
            L = int(rng.integers(1, S + 1))
            if name == "abstract_emb" and i % 7 == 3:
                L = 0  # a news without an abstract: all-masked view
            emb = rng.standard_normal((1, S, c["D"])).astype(np.float32)
            feats[name] = (emb, (np.arange(S)[None, :] < L).astype(np.float32))
        feats["category_index"] = int(rng.integers(1, 20))
        feats["subcategory_index"] = int(rng.integers(1, 301))
        news_feat[f"N{i}"] = feats
    ids = list(news_feat)
    sessions = []
    for s_i, (nh, npos, nneg) in enumerate(NAML_SESSIONS):
        pick = lambda k: [ids[int(j)] for j in rng.integers(0, len(ids), size=k)]  # noqa: E731
        sessions.append({"history": pick(nh), "positives": pick(npos), "negatives": pick(nneg),
                         "main_theme": f"theme{s_i % 3}", "main_category": "news", "user_index": s_i})
    return news_feat, sessions


def naml_data_cfg(c=None):
    c = c or NAML_DATA
    return model_cfg(dict(model="NAML", E=c["E"], bias=False, h=4, D=c["D"], H=c["l_hist"], S=c["S"]))


METRIC_CASES = {
    "plain": ([1, 0, 0, 1, 0, 0, 0], [0.9, 0.1, 0.5, 0.4, 0.45, 0.0, 0.3]),
    "relu_ties": ([1, 0, 0, 0, 1, 0], [0.7, 0.0, 0.0, 0.2, 0.0, 0.0]),
    "one_pos_last": ([0, 0, 0, 0, 1], [0.5, 0.4, 0.3, 0.2, 0.1]),
    "over_one": ([1, 0, 1, 0], [1.7, 0.6, 0.2, 0.49]),
    "twelve": ([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1], [0.31, 0.9, 0.12, 0.5, 0.77, 0.62, 0.05, 0.41, 0.8, 0.33, 0.2, 0.1]),
}
