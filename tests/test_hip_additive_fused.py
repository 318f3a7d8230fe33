"""GPU: the one-launch additive encoder (xnrs_amd/csrc/additive_fused.hip: fc1 + tanh + fc2 + exp * mask + normalise +
weighted sum of a TextEncoder without self-attention, layers.py:60-65 / news_encoding.py:48-59) against the GEMM + pooling
pipeline it replaces (BIT FOR BIT -- the dispatcher picks between them by batch size), the CPU oracle and the goldens of
the real reference."""
import numpy as np
import pytest
import torch

from oracle import xnrs_oracle as O
from tests import helpers as H
from tests.golden import cases
from xnrs_amd import hip, ops, synth
from xnrs_amd.models.components import layers, news_encoding

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def encoder(D, A, E, head, bias, seed):
    enc = news_encoding.TextEncoder(pooler=layers.AdditiveAttention(D, A), p_dropout=0.0, out_features=E if head else D,
                                    in_features=D, head=head, att=None, bias=bias)
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in enc.state_dict().items()}, seed)
    enc.load_state_dict(sd)
    return enc.eval().to(DEV), sd


def tokens(n, S, D, seed, full_pad_prob=0.2):
    rng = synth.rng_for(seed)
    x, m = synth.token_block(rng, 1, n, S, D, min_len=1, full_pad_prob=full_pad_prob)
    return x[0].to(DEV), m[0].to(DEV)


def run(enc, x, m, mode, ids=None, fbuf="1"):
    with hip.knobs(XNRS_ADDITIVE_FUSED=mode, XNRS_AF_FBUF=fbuf), torch.no_grad():
        return ops.text_encoder_forward(x, m, None, enc.pooler, getattr(enc, "head", None), ids=ids)


SHAPES = [  # n_news, S, D, A, head
    (7, 50, 768, 256, True),      # less than one tile... and a ragged last tile (5 news per tile)
    (1311, 50, 768, 256, True),   # several tiles per workgroup on a small grid? (263 tiles: > 256 CUs -> second round)
    (640, 50, 768, 256, False),   # exactly 128 tiles, no head
    (333, 30, 300, 256, True),    # 8 news per tile, K tail (300 = 18 * 16 + 12)
    (100, 64, 64, 200, False),    # 4 news per tile, A < 256 (zero-filled hidden columns), short contraction (4 K tiles)
    (77, 4, 16, 129, False),      # smallest eligible everything: 64 news per tile, ONE K tile (all pooling in the drain loop)
    (300, 25, 256, 256, True),    # the user-tower shape (H = 25, E = 256)
    (41, 13, 20, 130, False),     # D = 20: K tail inside the first chunk column
    (1000, 33, 772, 252, True),   # 7 news per tile, 3 pairs per thread... D % 16 = 4
    (513, 16, 1024, 256, False),  # 16 news per tile x 256 chunks = 4 096 pairs -> outside the plan (pipeline both times)
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_S%d_D%d_A%d_%s" % (s[0], s[1], s[2], s[3], "head" if s[4] else "nohead"))
def test_fused_equals_pipeline_bitwise_and_oracle(shape):
    n, S, D, A, head = shape
    enc, sd = encoder(D, A, 32, head, True, 1000 + n)
    x, m = tokens(n, S, D, 2000 + n)
    y0, hm0 = run(enc, x, m, "0")
    y1, hm1 = run(enc, x, m, "2")
    y2, hm2 = run(enc, x, m, "2", fbuf="2")
    assert torch.isfinite(y1).all()
    assert torch.equal(y1, y0) and torch.equal(hm1, hm0), (y1 - y0).abs().max().item()
    assert torch.equal(y2, y0) and torch.equal(hm2, hm0)
    yo, hmo = O.text_encoder(x.cpu()[None], m.cpu()[None], sd)
    H.assert_close(y1, yo[0], what="fused vs oracle")
    assert torch.equal(hm1.cpu(), hmo[0].reshape(-1))


def test_fused_id_gather_and_position_independence():
    """Rows gathered from a table by id == the materialised rows, and a news vector does not depend on where in the batch
    (which tile, which slot of the tile, which workgroup) the news sits."""
    S, D, A = 50, 768, 256
    enc, _ = encoder(D, A, 256, True, False, 77)
    tx, tm = tokens(900, S, D, 78)
    rng = np.random.default_rng(3)
    ids = torch.from_numpy(rng.integers(0, 900, size=1500).astype(np.int32)).to(DEV)
    y_ids, hm_ids = run(enc, tx, tm, "2", ids=ids)
    y_pipe, hm_pipe = run(enc, tx, tm, "0", ids=ids)
    y_dense, hm_dense = run(enc, tx[ids.long()].contiguous(), tm[ids.long()].contiguous(), "2")
    assert torch.equal(y_ids, y_pipe) and torch.equal(hm_ids, hm_pipe)
    assert torch.equal(y_ids, y_dense) and torch.equal(hm_ids, hm_dense)
    y_all, _ = run(enc, tx, tm, "2")
    assert torch.equal(y_ids, y_all[ids.long()])  # same news, other position / other batch: same bits
    one, _ = run(enc, tx[123:124].contiguous(), tm[123:124].contiguous(), "2")
    assert torch.equal(one[0], y_all[123])


@pytest.mark.parametrize("name", ["news_add_768", "news_add_head_nobias", "news_add_nohead"])
def test_fused_reproduces_the_reference_goldens(name):
    g = H.golden("encoders")
    c = cases.ENCODERS[name]
    enc = news_encoding.TextEncoder(pooler=layers.AdditiveAttention(c["D"], c["A"]), p_dropout=0.0, out_features=c["E"],
                                    in_features=c["D"], head=c["head"], att=None, bias=c["bias"])
    sd = synth.fill_state_dict({k: tuple(v.shape) for k, v in enc.state_dict().items()}, c["seed"] + 1)
    enc.load_state_dict(sd)
    enc = enc.eval().to(DEV)
    x, m = cases.encoder_inputs(c)
    with hip.knobs(XNRS_ADDITIVE_FUSED="2"), torch.no_grad():
        y, hm = enc((x.to(DEV), m.to(DEV)))
    H.assert_close(y, g[f"{name}/y"], what=name)
    assert torch.equal(hm.cpu(), torch.from_numpy(g[f"{name}/hm"]))


def test_default_dispatch_takes_the_fused_kernel_from_a_full_chip_on():
    """Default knob: >= 512 tiles (two per CU) -> the fused launch (profile stage 3 records ONE launch and no pooling
    launch), fewer -> the pipeline; the results agree bit for bit either way."""
    S, D, A = 50, 768, 256
    enc, _ = encoder(D, A, 256, True, False, 5)
    for n, fused in ((2555, False), (2556, True)):  # 511 and 512 tiles of 5 news
        x, m = tokens(n, S, D, 6, full_pad_prob=0.0)
        hip.profile_enable(hip.PROFILE_ALL)
        with torch.no_grad():
            y, _ = ops.text_encoder_forward(x, m, None, enc.pooler, enc.head)
        torch.cuda.synchronize()
        st = hip.profile_read()
        hip.profile_enable(0)
        if fused:
            assert st["pool"][1] == 0 and st["fc1_tanh_gemm"][1] == 1, (n, st)
        else:  # the pipeline: one fc1 GEMM and one pooling launch per 65 500-row pass
            assert st["pool"][1] == st["fc1_tanh_gemm"][1] >= 1, (n, st)
        y0, _ = run(enc, x, m, "0")
        assert torch.equal(y, y0)


def test_fast_tanh_knob():
    """XNRS_FAST_TANH=0 (ocml tanhf) and the default (hardware exp2 / rcp form) agree to ~1e-6 on the pooled vectors and
    both sit inside the parity bar; the fused kernel and the pipeline stay bitwise equal under either."""
    S, D, A = 50, 768, 256
    enc, sd = encoder(D, A, 256, False, True, 9)
    x, m = tokens(600, S, D, 10)
    yo, _ = O.text_encoder(x.cpu()[None], m.cpu()[None], sd)
    outs = {}
    for ft in ("1", "0"):
        with hip.knobs(XNRS_FAST_TANH=ft):
            y0, _ = run(enc, x, m, "0")
            y1, _ = run(enc, x, m, "2")
        assert torch.equal(y0, y1)
        H.assert_close(y1, yo[0], what=f"fast_tanh={ft}")
        outs[ft] = y1
    assert H.rel_err(outs["1"], outs["0"]) < 5e-6
