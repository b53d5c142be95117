"""CPU-side checks of the drop-in boundary: libses_hip.so loads, exports every symbol include/ses.h
declares, the ctypes table matches the header, and the product refuses to run without a GPU
(no CPU fallback).  No kernel is launched here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ses.h")

from ses import _lib  # noqa: E402  (simple-es_amd/ is on sys.path via conftest)


def header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = re.findall(r"(?:int|const char \*)\s*\**(ses_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)
    out = {}
    for name, args in decls:
        args = args.strip()
        out[name] = 0 if args in ("void", "") else len([a for a in args.split(",") if a.strip()])
    return out


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    assert os.path.dirname(_lib.LIB_PATH).endswith("simple-es_amd")


def test_every_declared_symbol_is_exported_and_bound():
    decl = header_functions()
    assert len(decl) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in include/ses.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name]) == nargs, f"{name}: header has {nargs} args, ctypes table {len(_lib.SIGNATURES[name])}"
    assert set(_lib.SIGNATURES) == set(decl)


def test_param_count_matches_reference_layout():
    lib = _lib.load()
    # P = 226 (CartPole MLP), 6756 (LunarLander GRU), 581 (simple_spread N=2), 773 (N=3)
    assert lib.ses_param_count(4, 2, 0) == 226
    assert lib.ses_param_count(8, 4, 1) == 6756
    assert lib.ses_param_count(12, 5, 0) == 581
    assert lib.ses_param_count(18, 5, 0) == 773
    assert lib.ses_param_count(4, 2, 1) == 6562
    assert lib.ses_version().startswith(b"ses-hip")


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    from ses import HipES, SesError
    with pytest.raises(SesError, match="no CPU fallback"):
        HipES()
    # the raw ABI refuses as well
    lib = _lib.load()
    cfg = _lib.SesConfig(0, 4, 2, 1, 0, 0, 500, 5, 0, 0, 1, 0)
    h = ctypes.c_void_p()
    rc = lib.ses_create(ctypes.byref(cfg), None, ctypes.byref(h))
    assert rc < 0 and lib.ses_last_error()


def test_product_never_imports_the_oracle():
    """The hot path must not route through oracle/ (tests, smoke() and bench's cpu_baseline only)."""
    src_root = os.path.join(ROOT, "simple-es_amd")
    offenders = []
    for d, _, files in os.walk(src_root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                text = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "ses_oracle" in text.replace(
                        "oracle/ses_oracle.c", ""):
                    offenders.append(os.path.join(d, f))
    assert not offenders, offenders


def test_every_tuning_knob_is_documented_in_the_header():
    """ses_set_tuning's table (csrc/ses_core.hip) and the knob list in include/ses.h's comment must not drift apart: every knob the
    library accepts is named in the header a caller reads."""
    import re
    core = open(os.path.join(ROOT, "simple-es_amd", "csrc", "ses_core.hip")).read()
    header = open(os.path.join(ROOT, "include", "ses.h")).read()
    knobs = re.findall(r'\{"([a-z0-9_]+)", &ses_handle::tune_', core)
    assert len(knobs) >= 30 and len(set(knobs)) == len(knobs)
    missing = [k for k in knobs if f'"{k}"' not in header]
    assert not missing, f"knobs accepted by ses_set_tuning but absent from include/ses.h: {missing}"
