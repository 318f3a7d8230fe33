"""CPU: xnrs_amd.install() -- the reference's import lines (train.py:12,18; `from ..components import layers` inside its
model files) resolve to the HIP-backed mirrors with no reference file touched.  Exercised against a STUB `xnrs` package
written to a temp dir (a few marker classes; nothing of the reference is copied), in a child interpreter so the mirrors
never leak into the test process; and, in the build container only, against the real reference tree."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(code, extra_path):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, extra_path]))
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=300)


def make_stub(tmp_path):
    pkg = tmp_path / "xnrs"
    for d in ("", "models", "models/components", "models/full_models"):
        (pkg / d).mkdir(parents=True, exist_ok=True)
    (pkg / "__init__.py").write_text("from .models import make_model as _mm  # the package pulls its models in, like the reference's\n")
    (pkg / "models" / "__init__.py").write_text("raise RuntimeError('the stub xnrs/models/__init__.py must never run after install()')\n")
    (pkg / "models" / "utils.py").write_text("def load_model_from_ckpt(p):\n    return ('stub-ckpt', p)\n\ndef get_checkpoint():\n    return 'stub'\n")
    (pkg / "models" / "make_model.py").write_text(
        "def make_model(cfg):\n    return ('stub-model', cfg.model)\n")
    (pkg / "models" / "components" / "__init__.py").write_text("raise RuntimeError('must not run')\n")
    (pkg / "models" / "components" / "layers.py").write_text(
        "class MaskedMax:\n    marker = 'stub-layers'\n\nclass AdditiveAttention:\n    marker = 'stub-must-lose'\n")
    (pkg / "models" / "components" / "news_encoding.py").write_text(
        "from . import layers\n\nclass CategoryEncoder:\n    uses = layers.MaskedMax\n")
    (pkg / "models" / "full_models" / "__init__.py").write_text("raise RuntimeError('must not run')\n")
    (pkg / "models" / "full_models" / "npa.py").write_text(
        "from ..components import layers, TextEncoder, CategoryEncoder\n\nclass NPA:\n    parts = (layers.MaskedMax, layers.AdditiveAttention, TextEncoder, CategoryEncoder)\n")
    return str(tmp_path)


def test_install_over_a_stub_package(tmp_path):
    stub = make_stub(tmp_path)
    r = run("""
        import xnrs_amd
        assert xnrs_amd.install() is True          # the stub package was found on sys.path (not imported)
        assert xnrs_amd.install() is True          # idempotent
        import xnrs                                 # the package's own __init__ imports .models -> the mirror
        from xnrs.models import make_model          # train.py:12
        from xnrs_amd.models import assemblies, blocks
        from xnrs_amd.models.components import layers as our_layers
        from xnrs.models.full_models import NRMS, NAML, NPA   # train.py:18: ours, ours, the stub's file
        assert NRMS is assemblies.NRMS and NAML is assemblies.NAML
        from xnrs.models.components import layers, scoring, TextEncoder, ParentRec, news_encoding
        assert TextEncoder is blocks.TextEncoder and ParentRec is blocks.ParentRec
        assert layers.AdditiveAttention is our_layers.AdditiveAttention and layers.MultiHeadAttention is our_layers.MultiHeadAttention
        assert layers.MaskedMax.marker == 'stub-layers'                       # not on the path: the package's own class
        assert NPA.parts[0] is layers.MaskedMax and NPA.parts[1] is our_layers.AdditiveAttention and NPA.parts[2] is blocks.TextEncoder
        assert NPA.parts[3].uses is layers.MaskedMax                           # components.CategoryEncoder from the package's file
        assert scoring.DotScoring.__module__.startswith('xnrs_amd')
        from xnrs.models import load_model_from_ckpt                          # re-export of xnrs/models/utils.py
        assert load_model_from_ckpt('p') == ('stub-ckpt', 'p')
        class Cfg(dict):
            __getattr__ = dict.__getitem__
        from xnrs_amd import synth
        m = make_model(Cfg(synth.model_cfg(dict(model='NRMS', E=16, bias=False, h=4, D=32, H=4, S=8))))
        assert type(m) is assemblies.NRMS
        assert make_model(Cfg(dict(synth.model_cfg(dict(model='NPA', E=16, bias=False, h=4, D=32, H=4, S=8))))) == ('stub-model', 'NPA')
        try:
            make_model(Cfg(dict(synth.model_cfg(dict(model='nope', E=16, bias=False, h=4, D=32, H=4, S=8)))))
            raise SystemExit('unknown model must raise')
        except ValueError:
            pass
        print('OK')
    """, stub)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_install_refuses_after_the_reference_models_were_imported(tmp_path):
    stub = make_stub(tmp_path)
    (tmp_path / "xnrs" / "models" / "__init__.py").write_text("make_model = 'reference'\n")
    (tmp_path / "xnrs" / "__init__.py").write_text("")
    r = run("""
        import xnrs.models
        import xnrs_amd
        try:
            xnrs_amd.install()
            raise SystemExit('must refuse')
        except RuntimeError as e:
            assert 'before' in str(e)
        print('OK')
    """, stub)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/xnrs"), reason="the reference tree exists in the build container only")
def test_install_over_the_real_reference_tree():
    """train.py:12 and :18 as committed, against /root/reference (the absent third-party packages that are not on the
    arithmetic path get empty stand-ins, SURVEY.md section 8c recipe 2)."""
    r = run("""
        import importlib.machinery, sys, types
        import transformers  # noqa: F401  (before the stand-ins, recipe 2)
        class DotMap(dict):
            __getattr__ = dict.__getitem__
        for name, attrs in [('dotmap', {'DotMap': DotMap}), ('omegaconf', {'DictConfig': dict}), ('wget', {}),
                            ('wandb', {'Histogram': lambda *a, **k: None, 'Table': lambda *a, **k: None, 'log': lambda *a, **k: None})]:
            mod = types.ModuleType(name); mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
            for k, v in attrs.items(): setattr(mod, k, v)
            sys.modules[name] = mod
        import xnrs_amd
        assert xnrs_amd.install() is True
        from xnrs.models import make_model
        from xnrs.data import make_mind_data
        from xnrs.training import BCELogitsRankingTrainer, MSERankingTrainer, ContrastiveRankingTrainer
        from xnrs.models.full_models import CAUM, LSTUR, NPA, NRMS, NAML, SmallNAML
        from xnrs_amd.models import assemblies
        assert NRMS is assemblies.NRMS and NAML is assemblies.NAML
        assert CAUM.__module__ == 'xnrs.models.full_models.caum' and NPA.__module__ == 'xnrs.models.full_models.npa'
        import yaml
        cfg = DotMap(yaml.safe_load(open('/root/reference/config/mind_small_NRMS.yml')))
        m = make_model(cfg)
        assert type(m) is assemblies.NRMS and sum(p.numel() for p in m.parameters()) == 3151364
        cfg = DotMap(yaml.safe_load(open('/root/reference/config/mind_small_NAML.yml')))
        assert type(make_model(cfg)) is assemblies.NAML
        print('OK')
    """, "/root/reference")
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
