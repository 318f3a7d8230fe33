"""Mirror of xnrs/models/make_model.py: the factory where the drop-in is selected."""
from .components import scoring
from .full_models import StandardRec, BaseRec, MeanRec, ParamFreeRec, NRMS, NAML

# reference models/scorers that are NOT on the path BASELINE.json names (SURVEY.md section 2 marks them out
# of scope): they keep running on the reference's own stock-torch classes.
_OUT_OF_SCOPE_MODELS = ('smallNAML', 'NPA', 'LSTUR', 'CAUM')
_OUT_OF_SCOPE_SCORING = ('bilin', 'fc', 'CAUMScoring')


def make_model(cfg):
    """xnrs/models/make_model.py:15-56.  Same cfg keys, same ValueError on unknown names."""
    if cfg.scoring == 'dot':
        scoring_fn = scoring.DotScoring()
    elif cfg.scoring in _OUT_OF_SCOPE_SCORING:
        raise NotImplementedError(
            f"cfg.scoring={cfg.scoring!r} is outside the MI355X hot path (dot-product scorer only); "
            "use the reference's torch implementation for it")
    else:
        # includes 'nonlin', whose class does not exist in the reference either (make_model.py:25-26)
        raise ValueError(f'invalid value for cfg.scoring: {cfg.scoring}')

    if cfg.model == 'standard':
        model = StandardRec(cfg, scoring_fn)
    elif cfg.model == 'base':
        model = BaseRec(cfg, scoring_fn)
    elif cfg.model == 'mean':
        model = MeanRec(cfg, scoring_fn)
    elif cfg.model == 'NRMS':
        model = NRMS(cfg, scoring_fn)
    elif cfg.model == 'NAML':
        model = NAML(cfg, scoring_fn)
    elif cfg.model in _OUT_OF_SCOPE_MODELS:
        raise NotImplementedError(
            f"cfg.model={cfg.model!r} is outside the MI355X hot path (NRMS / standard / base / mean / NAML); "
            "use the reference's torch implementation for it")
    else:
        raise ValueError(f'invalid value for cfg.model: {cfg.model}')
    return model
