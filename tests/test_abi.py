"""CPU: the C-ABI library loads and exports every symbol include/xnrs_hip.h declares; host-side
logic that needs no GPU (state_dict contract, factory errors, loud failure without a device)."""
import ctypes
import os
import re

import pytest
import torch

from tests import helpers as H
from tests.golden import cases
from xnrs_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Cfg(dict):
    __getattr__ = dict.__getitem__


def header_symbols():
    text = open(os.path.join(ROOT, "include", "xnrs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xnrs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 14
    assert sorted(hip.SYMBOLS) == syms, "xnrs_amd/hip.py SYMBOLS out of sync with include/xnrs_hip.h"
    l = ctypes.CDLL(hip.LIB_PATH)
    for s in syms:
        assert hasattr(l, s), f"libxnrs_hip.so does not export {s}"


def test_abi_version_and_error_strings():
    l = hip.lib()
    import re
    want = int(re.search(r"#define XNRS_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "xnrs_hip.h")).read()).group(1))
    assert l.xnrs_abi_version() == want == 6
    assert b"divisible" in l.xnrs_error_string(-2)
    assert l.xnrs_error_string(0) == b"ok"


def test_binary_was_built_from_the_sources_in_this_tree():
    """xnrs_build_id() = hash of csrc/*.hip + kernels.h + include/xnrs_hip.h at build time (csrc/Makefile); it must equal
    the hash of those files NOW -- a measured binary that is not the tree's fails here, without a rebuild."""
    assert hip.build_id() == hip.tree_build_id(), "libxnrs_hip.so is stale: run __graft_entry__.build()"
    assert len(hip.build_id()) == 16


def test_workspace_queries_are_pure_host_calls():
    l = hip.lib()
    n = l.xnrs_text_encoder_workspace_bytes(28160, 50, 768, 256, 256, 1, 0, 1, 0)
    assert 5e8 < n < 2e9  # chunked: ~64k rows per pass, not 28160*50 rows
    assert l.xnrs_text_encoder_workspace_bytes(10, 50, 768, 256, 256, 0, 1, 0, 0) == 0  # mean pool, no head
    assert l.xnrs_mha_workspace_bytes(4, 30, 300) >= 4 * 30 * 300 * 4 * 4


@pytest.mark.parametrize("name", sorted(cases.MODELS))
def test_state_dict_contract(name):
    """Key names and tensor shapes are part of the drop-in contract."""
    from xnrs_amd.models import make_model
    c = cases.MODELS[name]
    m = make_model(Cfg(cases.model_cfg(c)))
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v) for k, v in H.model_shapes(c).items()}


def test_nrms_param_count_matches_reference():
    from xnrs_amd.models import make_model
    cfg = Cfg(cases.model_cfg(dict(model="NRMS", E=256, bias=False, h=16, D=768, H=25, S=50)))
    m = make_model(cfg)
    assert sum(p.numel() for p in m.parameters()) == 3_151_364
    keys = list(m.state_dict().keys())
    # registration order of the reference MHA: q, v, k, out
    assert keys[1:9:2] == [f"news_encoder.att.{n}.weight" for n in ("q_linear", "v_linear", "k_linear", "out")]


def test_factory_errors():
    from xnrs_amd.models import make_model
    base = cases.model_cfg(cases.MODELS["nrms_tiny"])
    with pytest.raises(ValueError):
        make_model(Cfg({**base, "scoring": "nonlin"}))
    with pytest.raises(ValueError):
        make_model(Cfg({**base, "model": "nope"}))
    with pytest.raises(NotImplementedError):
        make_model(Cfg({**base, "model": "CAUM"}))


def test_no_silent_cpu_fallback():
    """Without a HIP device the product path must raise, never compute on the CPU."""
    from xnrs_amd.models import make_model
    c = cases.MODELS["nrms_tiny"]
    m = make_model(Cfg(cases.model_cfg(c))).eval()
    with torch.no_grad(), pytest.raises(hip.XnrsHipError):
        m(cases.model_batch(c))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "xnrs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("SURVEY", ""), f"{f} mentions the oracle"


def test_header_is_plain_c_and_cxx(tmp_path):
    """include/xnrs_hip.h is the drop-in boundary: it must compile as C99 and as C++ with nothing but the standard
    headers (no torch / HIP types in the signatures), and a C translation unit that takes the address of every
    declared entry point must compile against it."""
    import subprocess
    syms = header_symbols()
    src = tmp_path / "use.c"
    src.write_text('#include "xnrs_hip.h"\n#include <stddef.h>\n'
                   "const void *xnrs_all_entry_points[] = {\n" + "".join(f"  (const void *){s},\n" for s in syms) + "};\n"
                   "size_t xnrs_n_entry_points = sizeof(xnrs_all_entry_points) / sizeof(xnrs_all_entry_points[0]);\n")
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-Wno-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / "c.o")],
                ["g++", "-std=c++17", "-Wall", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "cxx.o")]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_graft_entry_build_agrees_with_the_header():
    """__graft_entry__.build() (the driver's "does it build" check) compiles -- a no-op when the library is current --
    imports the package and checks the library's ABI version against include/xnrs_hip.h; it must not carry a number
    of its own that a header bump leaves behind."""
    import __graft_entry__ as g
    g.build()
