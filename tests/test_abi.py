"""CPU: the C-ABI library builds, loads, and exports exactly what include/dctr.h declares."""
import ctypes
import os
import re

from helpers import GOLDEN_DIR  # noqa: F401  (path setup via conftest)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dctr.h")


def declared_functions(diag=False):
    """Entry points the header declares for the shipped library; ``diag=True``: the ones it declares only under
    DCTR_DIAG (the diagnostics build, libdctr_hip_diag.so)."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    blocks = re.findall(r"#ifdef DCTR_DIAG(.*?)#endif", src, flags=re.S)
    if diag:
        src = "\n".join(blocks)
    else:
        src = re.sub(r"#ifdef DCTR_DIAG.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dctr_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    names = declared_functions()
    for must in ("dctr_embed_fwd", "dctr_embed_bwd", "dctr_embed_apply", "dctr_fm_fwd", "dctr_fm_bwd"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from deepctr_torch._hip import lib as L
    assert L.available(), "run __graft_entry__.build() first"
    handle = ctypes.CDLL(L.LIB_PATH)
    for name in declared_functions():
        assert hasattr(handle, name), "libdctr_hip.so lacks %s declared in include/dctr.h" % name


def test_shipped_library_has_no_debug_hooks():
    """The dctr_dbg_* hooks and their mutable globals live only in the DCTR_DIAG build (round-1 verdict: the header
    promises a re-entrant library without global mutable state)."""
    from deepctr_torch._hip import lib as L
    handle = ctypes.CDLL(L.LIB_PATH)
    diag = declared_functions(diag=True)
    assert diag and sorted(L.DIAG_SIGNATURES) == diag
    for name in diag:
        assert not hasattr(handle, name), "%s must not be exported by the shipped library" % name


def test_binding_matches_header():
    from deepctr_torch._hip import lib as L
    assert sorted(set(L.SIGNATURES) - set(L.DIAG_SIGNATURES)) == declared_functions()
    lib = L.lib()  # loads, checks ABI version and struct sizes; no compute call
    assert lib.dctr_abi_version() == L.ABI_VERSION
    assert lib.dctr_sizeof_field() == ctypes.sizeof(L.Field) == 64
    assert lib.dctr_sizeof_plan() == ctypes.sizeof(L.Plan)
    assert b"invalid" in lib.dctr_strerror(-1)


def test_product_path_has_no_cpu_fallback():
    """A CPU tensor must raise, never silently compute."""
    import pytest
    import torch
    from deepctr_torch.inputs import SparseFeat
    from deepctr_torch.models import DeepFM
    cols = [SparseFeat("a", 5, 4), SparseFeat("b", 6, 4)]
    m = DeepFM(cols, cols, dnn_hidden_units=(4,), device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 2))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deepctr-torch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "np_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_cin_backward_workspace_covers_the_symmetric_layers_weight_staging():
    """dctr_cin_layer_bwd stages the folded weight slices of the symmetric first layer ([tiles][ot * 32][32] floats,
    k_cin_prep_wsym) in the caller's workspace before the partial sums go there: at small shapes that staging is the
    larger of the two (round 4: 5 fields, 8 feature maps, 64 samples wrote 3072 floats into 832).  Host-side arithmetic
    only -- callable without a GPU."""
    from deepctr_torch._hip import lib as L
    lib = L.lib()
    for (B, M, D, O) in [(64, 5, 4, 8), (16, 2, 4, 1), (1, 3, 8, 200), (4096, 26, 16, 128), (7, 32, 16, 33)]:
        ntiles = (M + 1) // 2 if M & 1 else (M + 1) // 2 + 1
        ot = (min(O, 128) + 31) // 32
        assert lib.dctr_cin_bwd_workspace_floats(B, M, M, D, O) >= ntiles * ot * 32 * 32, (B, M, D, O)
