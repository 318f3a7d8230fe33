"""CPU: the reference's shipped configs drop in unchanged (north_star).  tests/golden/shipped_configs.json was recorded by
tests/golden/make_golden.py from the REAL reference: for config/mind_small_{NRMS,CL,NAML,LSTUR}.yml the hyper-parameters
its model constructors read and the state_dict key -> shape map / parameter count of the reference's make_model(cfg).
xnrs_amd.make_model on the same keys must produce the same state_dict contract (checkpoints are exchanged by
state_dict: training.py:78-83, models/utils.py:17-20)."""
import json
import os

import pytest

from xnrs_amd.models import make_model

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "shipped_configs.json")))


class Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.mark.parametrize("name", ["mind_small_NRMS", "mind_small_CL", "mind_small_NAML"])
def test_shipped_config_builds_the_reference_state_dict(name):
    ref = FIX[name]
    model = make_model(Cfg(ref["cfg"]))
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(got) == list(ref["state_dict"]) or sorted(got) == sorted(ref["state_dict"])
    assert got == ref["state_dict"]
    assert sum(p.numel() for p in model.parameters()) == ref["n_params"]
    assert {"mind_small_NRMS": 3151364, "mind_small_CL": 656388, "mind_small_NAML": 1065494}[name] == ref["n_params"]


def test_lstur_config_is_refused_loudly():
    """config/mind_small_LSTUR.yml: the reference builds a GRU user tower that is outside the scoring hot path
    (SURVEY.md finding 5: only LSTUR's news encoder is in scope, xnrs_amd.models.LSTURNewsEncoder).  make_model says so
    instead of building something else."""
    ref = FIX["mind_small_LSTUR"]
    assert "state_dict" in ref  # the reference itself constructs it
    with pytest.raises(NotImplementedError, match="outside the MI355X hot path"):
        make_model(Cfg(ref["cfg"]))
