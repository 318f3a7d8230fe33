"""ctypes binding of libxnrs_hip.so (the C ABI declared in include/xnrs_hip.h).

There is deliberately NO fallback: if the shared library is missing, or a tensor is not on a HIP
device, these functions raise.  PyTorch is used only for device memory, streams and autograd
plumbing -- no torch op computes any part of the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
#: XNRS_HIP_LIB: another build of the same sources (diagnostic builds of tools/, e.g. libxnrs_hip_stamps.so); never a fallback
LIB_PATH = os.environ.get("XNRS_HIP_LIB") or os.path.join(_HERE, "libxnrs_hip.so")

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
POOL_ADDITIVE, POOL_MEAN = 0, 1

#: every symbol include/xnrs_hip.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = (
    "xnrs_abi_version", "xnrs_error_string", "xnrs_linear_fwd", "xnrs_mha_workspace_bytes", "xnrs_mha_fwd",
    "xnrs_additive_workspace_bytes", "xnrs_additive_attention_fwd", "xnrs_masked_mean_fwd", "xnrs_collapse_mask",
    "xnrs_text_encoder_workspace_bytes", "xnrs_text_encoder_fwd", "xnrs_user_encoder_workspace_bytes",
    "xnrs_user_encoder_fwd", "xnrs_dot_scoring_fwd", "xnrs_profile_enable", "xnrs_profile_read",
    "xnrs_set_gemm_mode", "xnrs_get_gemm_mode", "xnrs_reload_knobs",
    "xnrs_text_encoder_unpadded_workspace_bytes", "xnrs_text_encoder_fwd_unpadded",
    "xnrs_seq_encoder_saved_bytes", "xnrs_seq_encoder_fwd_train", "xnrs_seq_encoder_fwd_train_live",
    "xnrs_seq_encoder_bwd_workspace_bytes",
    "xnrs_seq_encoder_bwd", "xnrs_seq_encoder_bwd_live", "xnrs_linear_bwd_workspace_bytes", "xnrs_linear_bwd",
    "xnrs_embedding_linear_bwd_workspace_bytes", "xnrs_embedding_linear_bwd", "xnrs_dot_scoring_bwd", "xnrs_dot_scoring_norm_bwd",
    "xnrs_assemble_train_batch", "xnrs_assemble_eval_batch", "xnrs_score_csr", "xnrs_rank_metrics", "xnrs_gather_rows",
    "xnrs_infonce_saved_bytes", "xnrs_infonce_fwd", "xnrs_infonce_bwd", "xnrs_train_fold_enabled",
    "xnrs_fold_weights_workspace_bytes", "xnrs_fold_weights",
    "xnrs_text_encoder_compact_workspace_bytes", "xnrs_text_encoder_fwd_compact",
    "xnrs_seq_encoder_fwd_train_rows", "xnrs_seq_encoder_bwd_rows", "xnrs_seq_encoder_saved_qkv_offset",
    "xnrs_fold_head_weights_workspace_bytes", "xnrs_fold_head_weights",
    "xnrs_build_id", "xnrs_row_lists_workspace_bytes", "xnrs_build_row_lists", "xnrs_set_status_word", "xnrs_status_string",
)
POOL_NONE = -1
PROFILE_STAGES = ("qkv_gemm", "attention_core", "out_gemm", "fc1_tanh_gemm", "pool", "head_gemms", "news_fused",
                  "bwd_dw_gemms", "bwd_dx_gemms", "bwd_attention_core")
PROFILE_ALL = (1 << len(PROFILE_STAGES)) - 1


class XnrsHipError(RuntimeError):
    pass


class MhaParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo")] + [
        ("n_heads", C.c_int32), ("scaled", C.c_int32), ("dropout_p", C.c_float), ("seed", C.c_uint64), ("seed_dev", C.c_void_p)]


class AdditiveParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2")] + [("hidden", C.c_int32)] + [
        (n, C.c_void_p) for n in ("w1_folded", "b1_folded")]


class HeadParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w0", "b0", "w2", "b2")] + [("out_features", C.c_int32), ("activation", C.c_int32)] + [
        (n, C.c_void_p) for n in ("w0_folded", "b0_rowvec")]


class RowLists(C.Structure):
    """xnrs_row_lists: the unmasked token rows and the token rows of the non-empty news (include/xnrs_hip.h)."""
    _fields_ = [("live_rows", C.c_void_p), ("live_src_rows", C.c_void_p), ("n_live", C.c_int64),
                ("kv_rows", C.c_void_p), ("kv_src_rows", C.c_void_p), ("n_kv", C.c_int64), ("qkv_shared", C.c_void_p),
                ("counts_dev", C.c_void_p), ("dqkv_image", C.c_void_p), ("dqkv_mode", C.c_int32)]


DQKV_OWN, DQKV_DEFER, DQKV_MERGE = 0, 1, 2


class MhaGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo")]


class AdditiveGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2")]


class HeadGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w0", "b0", "w2", "b2")]


_lib = None


def lib():
    """Load libxnrs_hip.so once.  Missing library = hard error (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XnrsHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C xnrs_amd/csrc`).  xnrs_amd has no non-HIP fallback.")
    l = C.CDLL(LIB_PATH)
    p, i32, i64, sz, f = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float
    l.xnrs_abi_version.restype = i32
    l.xnrs_abi_version.argtypes = []
    l.xnrs_error_string.restype = C.c_char_p
    l.xnrs_error_string.argtypes = [i32]
    l.xnrs_linear_fwd.restype = i32
    l.xnrs_linear_fwd.argtypes = [p, p, i32, p, p, p, i64, i32, i32, i32, p]
    l.xnrs_mha_workspace_bytes.restype = sz
    l.xnrs_mha_workspace_bytes.argtypes = [i64, i32, i32]
    l.xnrs_mha_fwd.restype = i32
    l.xnrs_mha_fwd.argtypes = [p, p, C.POINTER(MhaParams), p, i64, i32, i32, p, sz, p]
    l.xnrs_additive_workspace_bytes.restype = sz
    l.xnrs_additive_workspace_bytes.argtypes = [i64, i32, i32, i32]
    l.xnrs_additive_attention_fwd.restype = i32
    l.xnrs_additive_attention_fwd.argtypes = [p, p, C.POINTER(AdditiveParams), p, p, i64, i32, i32, p, sz, p]
    l.xnrs_masked_mean_fwd.restype = i32
    l.xnrs_masked_mean_fwd.argtypes = [p, p, p, i64, i32, i32, p]
    l.xnrs_collapse_mask.restype = i32
    l.xnrs_collapse_mask.argtypes = [p, p, i64, i32, p]
    l.xnrs_text_encoder_workspace_bytes.restype = sz
    l.xnrs_text_encoder_workspace_bytes.argtypes = [i64, i32, i32, i32, i32, i32, i32, i32, i64]
    l.xnrs_text_encoder_fwd.restype = i32
    l.xnrs_text_encoder_fwd.argtypes = [p, p, p, i64, i32, i32, C.POINTER(MhaParams), i32, C.POINTER(AdditiveParams),
                                        C.POINTER(HeadParams), p, p, i64, p, sz, p]
    l.xnrs_text_encoder_unpadded_workspace_bytes.restype = sz
    l.xnrs_text_encoder_unpadded_workspace_bytes.argtypes = [i64, i64, i32, i32, i32, i32, i32, i32]
    l.xnrs_text_encoder_fwd_unpadded.restype = i32
    l.xnrs_text_encoder_fwd_unpadded.argtypes = [p, p, i64, i32, i32, p, p, i64, C.POINTER(MhaParams),
                                                 C.POINTER(AdditiveParams), C.POINTER(HeadParams), p, p, p, sz, p]
    l.xnrs_user_encoder_workspace_bytes.restype = sz
    l.xnrs_user_encoder_workspace_bytes.argtypes = [i64, i32, i32, i32, i32, i32, i32]
    l.xnrs_user_encoder_fwd.restype = i32
    l.xnrs_user_encoder_fwd.argtypes = [p, p, i64, i32, i32, C.POINTER(MhaParams), i32, C.POINTER(AdditiveParams),
                                        C.POINTER(HeadParams), p, p, p, sz, p]
    l.xnrs_dot_scoring_fwd.restype = i32
    l.xnrs_dot_scoring_fwd.argtypes = [p, p, p, i64, i32, i32, i32, p]
    l.xnrs_seq_encoder_saved_bytes.restype = sz
    l.xnrs_seq_encoder_saved_bytes.argtypes = [i64, i32, i32, i32, i32, i32, i32, i32]
    l.xnrs_seq_encoder_fwd_train.restype = i32
    l.xnrs_seq_encoder_fwd_train.argtypes = [p, p, p, i64, i32, i32, C.POINTER(MhaParams), i32, C.POINTER(AdditiveParams),
                                             C.POINTER(HeadParams), p, p, p, p, sz, p]
    l.xnrs_seq_encoder_fwd_train_live.restype = i32
    l.xnrs_seq_encoder_fwd_train_live.argtypes = [p, p, p, i64, i32, i32, C.POINTER(MhaParams), i32, C.POINTER(AdditiveParams),
                                                  C.POINTER(HeadParams), p, p, p, p, sz, p, p, i64, p]
    l.xnrs_seq_encoder_bwd_workspace_bytes.restype = sz
    l.xnrs_seq_encoder_bwd_workspace_bytes.argtypes = [i64, i32, i32, i32, i32, i32, i32, i32]
    l.xnrs_seq_encoder_bwd.restype = i32
    l.xnrs_seq_encoder_bwd.argtypes = [p, p, p, i64, i32, i32, C.POINTER(MhaParams), i32, C.POINTER(AdditiveParams),
                                       C.POINTER(HeadParams), p, sz, p, p, C.POINTER(MhaGrads), C.POINTER(AdditiveGrads),
                                       C.POINTER(HeadGrads), p, sz, p]
    l.xnrs_seq_encoder_bwd_live.restype = i32
    l.xnrs_seq_encoder_bwd_live.argtypes = [p, p, p, i64, i32, i32, C.POINTER(MhaParams), i32, C.POINTER(AdditiveParams),
                                            C.POINTER(HeadParams), p, sz, p, p, C.POINTER(MhaGrads), C.POINTER(AdditiveGrads),
                                            C.POINTER(HeadGrads), p, p, i64, p, sz, p]
    l.xnrs_seq_encoder_saved_qkv_offset.restype = sz
    l.xnrs_seq_encoder_saved_qkv_offset.argtypes = [i64, i32, i32, i32, i32, i32, i32, i32]
    l.xnrs_seq_encoder_fwd_train_rows.restype = i32
    l.xnrs_seq_encoder_fwd_train_rows.argtypes = [p, p, p, i64, i32, i32, C.POINTER(MhaParams), i32, C.POINTER(AdditiveParams),
                                                  C.POINTER(HeadParams), p, p, p, p, sz, C.POINTER(RowLists), p]
    l.xnrs_seq_encoder_bwd_rows.restype = i32
    l.xnrs_seq_encoder_bwd_rows.argtypes = [p, p, p, i64, i32, i32, C.POINTER(MhaParams), i32, C.POINTER(AdditiveParams),
                                            C.POINTER(HeadParams), p, sz, p, p, C.POINTER(MhaGrads), C.POINTER(AdditiveGrads),
                                            C.POINTER(HeadGrads), C.POINTER(RowLists), p, sz, p]
    l.xnrs_linear_bwd_workspace_bytes.restype = sz
    l.xnrs_linear_bwd_workspace_bytes.argtypes = [i64, i32, i32]
    l.xnrs_linear_bwd.restype = i32
    l.xnrs_linear_bwd.argtypes = [p, p, i32, p, p, p, p, p, i64, i32, i32, p, sz, p]
    l.xnrs_embedding_linear_bwd_workspace_bytes.restype = sz
    l.xnrs_embedding_linear_bwd_workspace_bytes.argtypes = [i64, i32, i32]
    l.xnrs_embedding_linear_bwd.restype = i32
    l.xnrs_embedding_linear_bwd.argtypes = [p, p, p, p, p, p, p, i64, i32, i32, i32, p, sz, p]
    l.xnrs_dot_scoring_bwd.restype = i32
    l.xnrs_dot_scoring_bwd.argtypes = [p, p, p, p, p, i64, i32, i32, p]
    l.xnrs_dot_scoring_norm_bwd.restype = i32
    l.xnrs_dot_scoring_norm_bwd.argtypes = [p, p, p, p, p, i64, i32, i32, p]
    l.xnrs_assemble_train_batch.restype = i32
    l.xnrs_assemble_train_batch.argtypes = [p, i64, p, p, p, p, p, p, i32, i32, i32, C.c_uint64, p, p, p]
    l.xnrs_assemble_eval_batch.restype = i32
    l.xnrs_assemble_eval_batch.argtypes = [p, i64, p, p, p, p, p, p, i32, i32, p, p, p, p, p, p]
    l.xnrs_gather_rows.restype = i32
    l.xnrs_gather_rows.argtypes = [p, p, p, i64, i64, p]
    l.xnrs_score_csr.restype = i32
    l.xnrs_score_csr.argtypes = [p, p, p, p, p, i64, i32, i32, p]
    l.xnrs_rank_metrics.restype = i32
    l.xnrs_rank_metrics.argtypes = [p, p, p, p, i64, p]
    l.xnrs_infonce_saved_bytes.restype = sz
    l.xnrs_infonce_saved_bytes.argtypes = [i64, i32]
    l.xnrs_infonce_fwd.restype = i32
    l.xnrs_infonce_fwd.argtypes = [p, p, i64, i32, f, p, p, sz, p]
    l.xnrs_infonce_bwd.restype = i32
    l.xnrs_infonce_bwd.argtypes = [p, i64, i32, f, p, sz, p, p, p]
    l.xnrs_set_gemm_mode.restype = i32
    l.xnrs_set_gemm_mode.argtypes = [i32]
    l.xnrs_get_gemm_mode.restype = i32
    l.xnrs_get_gemm_mode.argtypes = []
    l.xnrs_reload_knobs.restype = i32
    l.xnrs_reload_knobs.argtypes = []
    l.xnrs_profile_enable.restype = i32
    l.xnrs_profile_enable.argtypes = [C.c_uint32]
    l.xnrs_profile_read.restype = i32
    l.xnrs_profile_read.argtypes = [p, p, p]
    l.xnrs_train_fold_enabled.restype = i32
    l.xnrs_train_fold_enabled.argtypes = []
    l.xnrs_text_encoder_compact_workspace_bytes.restype = sz
    l.xnrs_text_encoder_compact_workspace_bytes.argtypes = [i64, i32, i32, i32, i32, i32, i32, i64]
    l.xnrs_text_encoder_fwd_compact.restype = i32
    l.xnrs_text_encoder_fwd_compact.argtypes = [p, p, p, i64, i32, i32, C.POINTER(MhaParams), C.POINTER(AdditiveParams),
                                                C.POINTER(HeadParams), p, p, i64, p, sz, p]
    l.xnrs_fold_weights_workspace_bytes.restype = sz
    l.xnrs_fold_weights_workspace_bytes.argtypes = [i32, i32]
    l.xnrs_fold_weights.restype = i32
    l.xnrs_fold_weights.argtypes = [C.POINTER(MhaParams), C.POINTER(AdditiveParams), i32, p, p, p, sz, p]
    l.xnrs_build_id.restype = C.c_char_p
    l.xnrs_build_id.argtypes = []
    l.xnrs_row_lists_workspace_bytes.restype = sz
    l.xnrs_row_lists_workspace_bytes.argtypes = [i64]
    l.xnrs_build_row_lists.restype = i32
    l.xnrs_build_row_lists.argtypes = [p, p, i64, i32, p, p, p, p, p, p, sz, p]
    l.xnrs_fold_head_weights_workspace_bytes.restype = sz
    l.xnrs_fold_head_weights_workspace_bytes.argtypes = [i32, i32]
    l.xnrs_fold_head_weights.restype = i32
    l.xnrs_fold_head_weights.argtypes = [C.POINTER(MhaParams), C.POINTER(HeadParams), i32, p, p, p, sz, p]
    l.xnrs_set_status_word.restype = i32
    l.xnrs_set_status_word.argtypes = [p]
    l.xnrs_status_string.restype = C.c_char_p
    l.xnrs_status_string.argtypes = [i32]
    if l.xnrs_abi_version() != 6:
        raise XnrsHipError("libxnrs_hip.so ABI version mismatch; rebuild it")
    _lib = l
    return l


# ---- sticky device status word (include/xnrs_hip.h): what the sync-free entry points could not raise
STATUS_NONBINARY_MASK, STATUS_ROW_RANGE = 1, 2
_status = {}


def status_word(device) -> torch.Tensor:
    """The int32 device word the library's sync-free paths OR their error bits into (one per process: registered with the
    library on first use; xnrs_amd runs one device per process).  Read it with check_status() at a natural sync point."""
    dev = torch.device(device)
    w = _status.get(dev.index)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=dev)
        if not _status:  # the library holds ONE pointer, valid on the device current at registration (its launches on another
            # device ignore it): the first device's word is the registered one; further devices get a host-layer-only
            # word -- the row-range bit of NewsStore.gather still lands there, the library's own bits stay NaN-only
            with torch.cuda.device(dev):
                check(lib().xnrs_set_status_word(C.c_void_p(w.data_ptr())), "xnrs_set_status_word")
        _status[dev.index] = w
    return w


def check_status(device=None):
    """Read (one host sync) and clear the status word(s); raise XnrsHipError for what a sync-free call could not report:
    a non-binary mask in the device-compacted encoder (its outputs are NaN), a table row id outside the table."""
    for idx, w in list(_status.items()):
        if device is not None and torch.device(device).index != idx:
            continue
        v = int(w.item())
        if v:
            w.zero_()
            raise XnrsHipError(f"device status word {v}: {lib().xnrs_status_string(v).decode()}")


def clear_status():
    """Zero the status word(s) without reading them (no sync)."""
    for w in _status.values():
        w.zero_()


def build_id() -> str:
    """Hash of the sources libxnrs_hip.so was built from (include/xnrs_hip.h: xnrs_build_id)."""
    return lib().xnrs_build_id().decode()


def tree_build_id() -> str:
    """The same hash over the sources in THIS tree (xnrs_amd/csrc/Makefile: ID_SRCS): equal to build_id() iff the binary
    was built from them."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")), key=os.path.basename)
    files += [os.path.join(csrc, "kernels.h"), os.path.join(os.path.dirname(_HERE), "include", "xnrs_hip.h")]
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(rc: int, what: str):
    if rc == 0:
        return
    msg = lib().xnrs_error_string(rc).decode()
    if rc == -2:
        # same exception type as the reference's failing .view() (layers.py:111,133)
        raise RuntimeError(f"{what}: {msg}")
    raise XnrsHipError(f"{what}: {msg} (code {rc})")


GEMM_F32, GEMM_BF16X3, GEMM_BF16X2 = 0, 1, 2


def set_gemm_mode(mode: int) -> int:
    """Arithmetic of the forward GEMMs (include/xnrs_hip.h: xnrs_set_gemm_mode); returns the previous mode."""
    return lib().xnrs_set_gemm_mode(int(mode))


def get_gemm_mode() -> int:
    return lib().xnrs_get_gemm_mode()


class knobs:
    """Development knobs for A/B runs and tests: ``with hip.knobs(XNRS_GEMM_PIPE="1"): ...`` sets the environment
    variables, makes the library re-read them (it reads them only at load and on xnrs_reload_knobs), and restores
    both on exit.  A value of None removes the variable."""

    def __init__(self, **env):
        self.env = env
        self.old = {}

    def __enter__(self):
        for k, v in self.env.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        lib().xnrs_reload_knobs()
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib().xnrs_reload_knobs()
        return False


def reload_knobs():
    lib().xnrs_reload_knobs()


def profile_enable(stage_mask: int):
    check(lib().xnrs_profile_enable(stage_mask), "xnrs_profile_enable")


def profile_read():
    """-> {stage: (ms, launches, flops)} of the launches recorded since profile_enable()."""
    n = len(PROFILE_STAGES)
    ms, ln, fl = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
    check(lib().xnrs_profile_read(ms, ln, fl), "xnrs_profile_read")
    return {PROFILE_STAGES[i]: (ms[i], ln[i], fl[i]) for i in range(n)}


# ---------------------------------------------------------------------------------- tensor plumbing
def dev_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    """Contiguous fp32 HIP tensor or a loud error (never a silent CPU path)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise XnrsHipError(f"{what}: tensor is on {t.device}; xnrs_amd runs on a HIP device only "
                           "(move the module and inputs to 'cuda'; there is no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_workspaces = {}


def workspace(device, nbytes: int) -> Optional[torch.Tensor]:
    """Grow-only per-(device, stream) scratch buffer handed to the library (it never allocates)."""
    if nbytes == 0:
        return None
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _workspaces.pop(key, None)
        buf = torch.empty(int(nbytes * 1.05) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def release_workspaces():
    _workspaces.clear()


# ---------------------------------------------------------------------------------- parameter packs
def mha_params(att, dropout_p: float = 0.0, seed: int = 0):
    """(MhaParams, keepalive list) from an xnrs_amd MultiHeadAttention module."""
    ts = [dev_f32(t, "mha weight") for t in (
        att.q_linear.weight, att.q_linear.bias, att.k_linear.weight, att.k_linear.bias,
        att.v_linear.weight, att.v_linear.bias, att.out.weight, att.out.bias)]
    p = MhaParams(*[t.data_ptr() for t in ts], att.h, 1 if att.scaled else 0, float(dropout_p), int(seed))
    return p, ts


def additive_params(pool, att=None):
    """att (optional, inference): the attention stage the pooler sits behind in this call -- the folded fc1 (W1.Wo,
    W1.bo + b1; include/xnrs_hip.h: xnrs_additive_params.w1_folded) is then attached from a per-module cache instead of
    being rebuilt by every call."""
    ts = [dev_f32(t, "additive weight") for t in (pool.fc1.weight, pool.fc1.bias, pool.fc2.weight, pool.fc2.bias)]
    p = AdditiveParams(*[t.data_ptr() for t in ts], pool.fc1.out_features, None, None)
    if att is not None and FOLD_CACHE:
        w1f, b1f = folded_fc1(att, pool)
        p.w1_folded, p.b1_folded = w1f.data_ptr(), b1f.data_ptr()
        ts = ts + [w1f, b1f]
    return p, ts


#: cache of folded fc1 weights per (attention module, pooler module): {key: (weakrefs of the four source tensors, versions,
#: w1f, b1f)}.  The C ABI keeps no state between calls; this is the CALLER's cache it is designed for.  A hit needs the four
#: source tensors to be the very same objects (weak references, compared with `is`: a rebuilt model whose tensors land on
#: recycled addresses can never match) at the same version counter.  An optimizer step, load_state_dict and every other
#: in-place write through the tensor bump the counter; a write through `p.data` does NOT (torch gives `.data` its own
#: counter): after `p.data.copy_(...)` / `dist.broadcast(p.data, ...)` call invalidate_fold_cache()
#: (xnrs_amd.distributed.broadcast_parameters does, and writes through `p.detach()`, which shares the counter).
FOLD_CACHE = os.environ.get("XNRS_FOLD_CACHE", "1") != "0"
_fold_cache = {}


def invalidate_fold_cache():
    """Forget every cached folded fc1 pair (and the grad step's shared projections / outputs): call after writing weights in
    a way torch's version counters do not see (`p.data.copy_`, a collective on `p.data`, a raw pointer write)."""
    _fold_cache.clear()
    from . import autograd
    autograd._QKV_IMAGES.clear()
    autograd._OUTPUTS.clear()
    autograd._TRAIN_FOLDS.clear()


def folded_fc1(att, pool):
    import weakref
    src = (att.out.weight, att.out.bias, pool.fc1.weight, pool.fc1.bias)
    ver = tuple(None if t is None else (t._version, t.data_ptr(), str(t.device), t.dtype) for t in src)
    key = (id(att), id(pool))
    hit = _fold_cache.get(key)
    if hit is not None and hit[1] == ver and all((r is None and t is None) or (r is not None and r() is t)
                                                 for r, t in zip(hit[0], src)):
        return hit[2], hit[3]
    ap, keep_a = mha_params(att)
    pp, keep_p = additive_params(pool)
    dev = pool.fc1.weight.device
    A, D = pool.fc1.out_features, pool.fc1.in_features
    w1f = torch.empty((A, D), dtype=torch.float32, device=dev)
    b1f = torch.empty((A,), dtype=torch.float32, device=dev)
    l = lib()
    nws = l.xnrs_fold_weights_workspace_bytes(D, A)
    ws = workspace(dev, nws)
    check(l.xnrs_fold_weights(C.byref(ap), C.byref(pp), D, ptr(w1f), ptr(b1f), ptr(ws), nws, stream_ptr(dev)), "xnrs_fold_weights")
    if len(_fold_cache) > 64:  # modules come and go (tests): keep the table small
        _fold_cache.clear()
    _fold_cache[key] = (tuple(None if t is None else weakref.ref(t) for t in src), ver, w1f, b1f)
    return w1f, b1f


def head_params(head, att=None):
    """nn.Sequential(Linear, ReLU, Linear) -> (HeadParams, keepalive list); biases may be absent.
    att (optional, inference with an additive pooler behind this attention stage): the head's first layer folded behind the
    out-projection (include/xnrs_hip.h: xnrs_head_params.w0_folded) rides along -- from the per-module cache, or computed
    for this call when XNRS_FOLD_CACHE=0 (the same kernels: the same bits)."""
    l0, l2 = head[0], head[2]
    ts = [dev_f32(l0.weight, "head.0.weight"), None if l0.bias is None else dev_f32(l0.bias, "head.0.bias"),
          dev_f32(l2.weight, "head.2.weight"), None if l2.bias is None else dev_f32(l2.bias, "head.2.bias")]
    hp = HeadParams(*[None if t is None else t.data_ptr() for t in ts], l0.out_features, head_activation(head[1]), None, None)
    if att is not None and FOLD_HEAD:
        w0f, b0v = folded_head(att, head)
        hp.w0_folded = w0f.data_ptr()
        hp.b0_rowvec = None if b0v is None else b0v.data_ptr()
        ts = ts + [w0f, b0v]
    return hp, ts


#: fold the attention out-projection into the head's first layer in inference (XNRS_FOLD_HEAD=0: off; XNRS_FOLD_OUT=0 turns
#: the whole fold off in the library, and with it this one)
FOLD_HEAD = os.environ.get("XNRS_FOLD_HEAD", "1") != "0" and os.environ.get("XNRS_FOLD_OUT", "1") != "0"


def folded_head(att, head):
    """(W0 . Wo [E, D], W0 . bo [E] or None) of an (attention stage, head) pair, cached like folded_fc1."""
    import weakref
    src = (att.out.weight, att.out.bias, head[0].weight)
    ver = tuple(None if t is None else (t._version, t.data_ptr(), str(t.device), t.dtype) for t in src)
    key = (id(att), id(head), "head")
    hit = _fold_cache.get(key) if FOLD_CACHE else None
    if hit is not None and hit[1] == ver and all((r is None and t is None) or (r is not None and r() is t)
                                                 for r, t in zip(hit[0], src)):
        return hit[2], hit[3]
    ap, keep_a = mha_params(att)
    l0, l2 = head[0], head[2]
    ts = [dev_f32(l0.weight, "head.0.weight"), None, dev_f32(l2.weight, "head.2.weight"), None]
    hp = HeadParams(ts[0].data_ptr(), None, ts[2].data_ptr(), None, l0.out_features, ACT_NONE, None, None)
    dev = l0.weight.device
    E, D = l0.out_features, l0.in_features
    w0f = torch.empty((E, D), dtype=torch.float32, device=dev)
    b0v = torch.empty((E,), dtype=torch.float32, device=dev) if att.out.bias is not None else None
    l = lib()
    nws = l.xnrs_fold_head_weights_workspace_bytes(D, E)
    ws = workspace(dev, nws)
    check(l.xnrs_fold_head_weights(C.byref(ap), C.byref(hp), D, ptr(w0f), ptr(b0v), ptr(ws), nws, stream_ptr(dev)),
          "xnrs_fold_head_weights")
    if FOLD_CACHE:
        if len(_fold_cache) > 64:
            _fold_cache.clear()
        _fold_cache[key] = (tuple(None if t is None else weakref.ref(t) for t in src), ver, w0f, b0v)
    return w0f, b0v


def head_activation(mod) -> int:
    """The head's activation module -> epilogue code.  nn.ReLU (the reference default, news_encoding.py:18),
    nn.Tanh and nn.Identity are what the GEMM epilogue and its backward implement; anything else is refused."""
    import torch.nn as nn
    if isinstance(mod, nn.ReLU):
        return ACT_RELU
    if isinstance(mod, nn.Tanh):
        return ACT_TANH
    if isinstance(mod, nn.Identity):
        return ACT_NONE
    raise NotImplementedError(f"head activation {type(mod).__name__}: the HIP head implements nn.ReLU (reference default), "
                              "nn.Tanh and nn.Identity")
