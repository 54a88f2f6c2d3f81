"""The C-ABI library loads on a CPU-only host, exports every symbol include/commpy_amd.h declares,
and the product path fails loudly (no CPU fallback) when no HIP device is present."""
import os
import re

import numpy as np
import pytest

from commpy_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "commpy_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(cpx_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _header_symbols()
    assert declared, "no declarations found in include/commpy_amd.h"
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert lib.cpx_version() >= 100


def test_no_cpu_fallback_without_device():
    """On a box without a GPU every compute entry point must raise, never compute on the host."""
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    from commpy_amd.channelcoding import Trellis, ldpc_bp_decode, map_decode, viterbi_decode
    from commpy_amd.modulation import QAMModem
    tr = Trellis(np.array([2]), np.array([[5, 7]]))
    with pytest.raises(_lib.EngineError):
        viterbi_decode(np.zeros(20), tr)
    with pytest.raises(_lib.EngineError):
        map_decode(np.zeros(8), np.zeros(8), tr, 1.0, np.zeros(8))
    with pytest.raises(_lib.EngineError):
        QAMModem(4).demodulate(np.zeros(4, complex), "soft", 1.0)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import ldpc_params
    with pytest.raises(_lib.EngineError):
        ldpc_bp_decode(np.ones(96), ldpc_params("gallager96"), "MSA", 2)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: no file under commpy_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "commpy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "cpx_oracle" not in src and "libcpx_oracle" not in src, f
