"""`xnrs_amd.install()`: make the reference's own import lines resolve to the HIP-backed modules, with the reference's
files untouched.

    import xnrs_amd; xnrs_amd.install()        # before anything imports `xnrs`  (e.g. in sitecustomize / the job script)
    python train.py --config config/mind_small_NRMS.yml     # /root/reference/train.py:12,18 as committed

What it does: registers import-path mirrors in `sys.modules` BEFORE the reference package is imported --

    xnrs.models                              make_model (ours; reference fallback for models outside the hot path)
    xnrs.models.components                   TextEncoder, UserEncoder, ParentRec (ours) + the submodules below
    xnrs.models.components.layers            AdditiveAttention, MultiHeadAttention, MaskedMean (ours); every other name
    xnrs.models.components.news_encoding       (MaskedMax, PersonalizedAttention, CategoryEncoder, ... of NPA / CAUM) is
    xnrs.models.components.user_encoding       looked up in the REFERENCE's own file, loaded lazily under a private name
    xnrs.models.components.scoring
    xnrs.models.components.parent
    xnrs.models.full_models                  NRMS, NAML, StandardRec, BaseRec, MeanRec, ParamFreeRec (ours); CAUM, LSTUR,
                                             NPA, SmallNAML from the reference's files (they run on stock torch)

-- so `from xnrs.models import make_model` (train.py:12), `from xnrs.models.full_models import CAUM, LSTUR, NPA, NRMS,
NAML, SmallNAML` (train.py:18), `from .models.utils import ...` (explain.py) and `from ..components import layers`
inside the reference's own model files all keep working.  The mirrors carry the reference package's directories in
`__path__` (found with importlib WITHOUT importing `xnrs`), which is how un-mirrored submodules (`xnrs.models.utils`,
`xnrs.models.full_models.caum`, ...) still load from the reference.  Nothing of the reference is copied or edited.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types
from typing import Dict, Optional

_MIRRORED = ("xnrs.models", "xnrs.models.components", "xnrs.models.components.layers", "xnrs.models.components.news_encoding",
             "xnrs.models.components.user_encoding", "xnrs.models.components.scoring", "xnrs.models.components.parent",
             "xnrs.models.full_models", "xnrs.models.make_model")
_state: Dict[str, object] = {"installed": False}


def _ref_root() -> Optional[str]:
    """Directory of the reference's `xnrs` package, found without importing it (None: not on sys.path)."""
    try:
        spec = importlib.util.find_spec("xnrs")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.submodule_search_locations:
        return None
    return list(spec.submodule_search_locations)[0]


def _load_private(alias: str, path: str, package: str):
    """Execute a reference source file under a private module name (its relative imports resolve inside `package`, i.e.
    through the mirrors).  Cached in sys.modules."""
    if alias in sys.modules:
        return sys.modules[alias]
    spec = importlib.util.spec_from_file_location(alias, path)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = package
    sys.modules[alias] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(alias, None)
        raise
    return mod


class _Mirror(types.ModuleType):
    """A module whose own attributes (the HIP-backed classes) win, and whose missing names are looked up in the
    reference's file of the same import path, loaded on first use."""

    def __init__(self, name: str, ours: Dict[str, object], ref_file: Optional[str], package: str, path=None, doc: str = ""):
        super().__init__(name, doc)
        self.__dict__.update(ours)
        self.__package__ = package
        if path is not None:
            self.__path__ = path
        self._xnrs_amd_ref_file = ref_file
        self._xnrs_amd_mirror = True

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        ref_file = self.__dict__.get("_xnrs_amd_ref_file")
        if ref_file and os.path.exists(ref_file):
            ref = _load_private(self.__name__ + "._reference", ref_file, self.__dict__["__package__"])
            if hasattr(ref, item):
                return getattr(ref, item)
        raise AttributeError(f"module {self.__name__!r} (xnrs_amd mirror) has no attribute {item!r}")


def install(force: bool = False) -> bool:
    """Register the mirrors.  Must run before `xnrs.models` is imported (raises otherwise unless it already IS the mirror);
    idempotent.  Returns True when the reference package was found on sys.path (False: mirrors only)."""
    if _state["installed"] and not force:
        return bool(_state["ref"])
    present = sys.modules.get("xnrs.models")
    if present is not None and not getattr(present, "_xnrs_amd_mirror", False):
        raise RuntimeError("xnrs_amd.install() must run before the reference's xnrs.models is imported "
                           "(call it first thing in the job script / sitecustomize)")
    from . import models as M
    from .models import assemblies, blocks
    from .models.components import layers as L, news_encoding as NE, parent as P, scoring as SC, user_encoding as UE

    root = _ref_root()
    mdir = os.path.join(root, "models") if root else None
    cdir = os.path.join(mdir, "components") if mdir else None
    fdir = os.path.join(mdir, "full_models") if mdir else None

    def ref(d, f):
        return os.path.join(d, f) if d else None

    def names(mod):
        return {k: v for k, v in vars(mod).items() if not k.startswith("_")}

    comp_sub = {
        "layers": _Mirror("xnrs.models.components.layers", names(L), ref(cdir, "layers.py"), "xnrs.models.components"),
        "news_encoding": _Mirror("xnrs.models.components.news_encoding", names(NE), ref(cdir, "news_encoding.py"), "xnrs.models.components"),
        "user_encoding": _Mirror("xnrs.models.components.user_encoding", names(UE), ref(cdir, "user_encoding.py"), "xnrs.models.components"),
        "scoring": _Mirror("xnrs.models.components.scoring", names(SC), ref(cdir, "scoring.py"), "xnrs.models.components"),
        "parent": _Mirror("xnrs.models.components.parent", names(P), ref(cdir, "parent.py"), "xnrs.models.components"),
    }
    components = _Mirror("xnrs.models.components",
                         dict(TextEncoder=blocks.TextEncoder, UserEncoder=blocks.UserEncoder, ParentRec=blocks.ParentRec, **comp_sub),
                         None, "xnrs.models.components", path=[cdir] if cdir else [])
    # names of xnrs/models/components/__init__.py that live in the reference only (CategoryEncoder of NPA / CAUM)
    components.__dict__["_xnrs_amd_ref_file"] = None
    comp_fallback = {"CategoryEncoder": "news_encoding"}

    def comp_getattr(item, _c=components, _fb=comp_fallback):
        if item in _fb:
            return getattr(_c.__dict__[_fb[item]], item)
        raise AttributeError(f"module 'xnrs.models.components' (xnrs_amd mirror) has no attribute {item!r}")
    components.__class__ = type("_Components", (_Mirror,), {"__getattr__": lambda self, item: comp_getattr(item)})

    ours_full = dict(NRMS=assemblies.NRMS, NRMS_LF=assemblies.NRMS_LF, NAML=assemblies.NAML, StandardRec=assemblies.StandardRec,
                     BaseRec=assemblies.BaseRec, MeanRec=assemblies.MeanRec, ParamFreeRec=assemblies.ParamFreeRec,
                     LSTURNewsEncoder=assemblies.LSTURNewsEncoder)
    # out-of-scope models stay the reference's classes, from the reference's files: name -> (file, attribute)
    ref_models = {"CAUM": "caum", "LSTUR": "lstur", "NPA": "npa", "SmallNAML": "naml"}
    full = _Mirror("xnrs.models.full_models", ours_full, None, "xnrs.models.full_models", path=[fdir] if fdir else [])

    def full_getattr(item):
        if item in ref_models and fdir:
            mod = importlib.import_module("xnrs.models.full_models." + ref_models[item])  # the reference's file, via __path__
            return getattr(mod, item)
        raise AttributeError(f"module 'xnrs.models.full_models' (xnrs_amd mirror) has no attribute {item!r}")
    full.__class__ = type("_FullModels", (_Mirror,), {"__getattr__": lambda self, item: full_getattr(item)})

    def make_model(cfg):
        """xnrs/models/make_model.py:15-56 on the HIP path; a model or scorer outside the path (NPA, CAUM, LSTUR, bilinear /
        fc scoring) is built by the REFERENCE's own make_model on stock torch, exactly as before the install."""
        try:
            return assemblies.make_model(cfg)
        except NotImplementedError:
            if not mdir:
                raise
            refmm = _load_private("xnrs.models.make_model._reference", os.path.join(mdir, "make_model.py"), "xnrs.models")
            return refmm.make_model(cfg)

    mm = _Mirror("xnrs.models.make_model", dict(make_model=make_model), None, "xnrs.models")
    models = _Mirror("xnrs.models", dict(make_model=make_model, components=components, full_models=full),
                     None, "xnrs.models", path=[mdir] if mdir else [])

    def models_getattr(item):
        if item in ("get_checkpoint", "load_model_from_ckpt") and mdir:  # xnrs/models/__init__.py re-exports them from utils
            return getattr(importlib.import_module("xnrs.models.utils"), item)
        if item == "utils" and mdir:
            return importlib.import_module("xnrs.models.utils")
        raise AttributeError(f"module 'xnrs.models' (xnrs_amd mirror) has no attribute {item!r}")
    models.__class__ = type("_Models", (_Mirror,), {"__getattr__": lambda self, item: models_getattr(item)})

    sys.modules["xnrs.models"] = models
    sys.modules["xnrs.models.make_model"] = mm
    sys.modules["xnrs.models.components"] = components
    for k, v in comp_sub.items():
        sys.modules["xnrs.models.components." + k] = v
    sys.modules["xnrs.models.full_models"] = full
    _state.update(installed=True, ref=root)
    del M
    return bool(root)


def uninstall() -> None:
    """Remove the mirrors (tests)."""
    for k in list(sys.modules):
        if k in _MIRRORED or k.endswith("._reference") and k.startswith("xnrs.models"):
            if getattr(sys.modules[k], "_xnrs_amd_mirror", False) or k.endswith("._reference"):
                sys.modules.pop(k, None)
    _state.update(installed=False, ref=None)
