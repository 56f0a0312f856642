"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol include/f16_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_symbols():
    src = open(os.path.join(REPO, "include", "f16_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_library_exports_every_declared_symbol():
    from f16_mpc_oop_py_amd import lib
    so = lib.build()
    L = ctypes.CDLL(so)
    syms = declared_symbols()
    assert "Nlplant" in syms and "atmos" in syms and "f16_rollout" in syms and len(syms) >= 15
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_python_constants_equal_the_header():
    """Status bits and behaviour flags of include/f16_hip.h, as the Python binding spells them."""
    from f16_mpc_oop_py_amd import lib
    src = open(os.path.join(REPO, "include", "f16_hip.h")).read()
    val = lambda name: eval(re.search(r"#define\s+%s\s+(\(?[0-9a-fx<>() u]+\)?)" % name, src).group(1).replace("u", ""))
    for k, v in lib.F16_ST.items():
        assert val("F16_ST_" + k) == v, k
    assert val("F16_ST_LOOP_STALL") == lib.F16_ST_LOOP_STALL and lib.F16_ST_ENV_STATE(3) == 1 << 11
    for k in ("FIX_CLR", "NO_ENVELOPE", "ONE_LANE", "HOLD_COMMAND"):
        assert val("F16_FLAG_" + k) == getattr(lib, "F16_FLAG_" + k), k


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from f16_mpc_oop_py_amd import F16Batch, lib
    with pytest.raises(lib.F16HipError):
        F16Batch([[0.0] * 18])


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(REPO, "f16_mpc_oop_py_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                for needle in ("import oracle", "from oracle", "oracle/", "libf16_oracle", "f16o_"):
                    assert needle not in txt, (root, f, needle)
