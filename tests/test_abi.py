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
                # oracle/_ref (the reference's own files, shipped for bench.py's cpu_baseline) and the reference checkout
                # are just as far out of the product's reach: no import of `commpy` proper, no path into either
                assert not re.search(r"^\s*(import|from)\s+commpy(\.|\s)", src, flags=re.M), f
                assert "oracle/_ref" not in src and "make_ref" not in src and "CPX_REFERENCE_PATH" not in src, f


def test_new_entry_points_fail_loudly_without_device():
    if _lib.device_count() > 0:
        pytest.skip("a HIP device is present")
    from commpy_amd.channelcoding import Trellis
    from commpy_amd.modulation import QAMModem
    from commpy_amd.parallel import DeviceGroup, RankComm
    tr = Trellis(np.array([2]), np.array([[5, 7]]))
    with pytest.raises(_lib.EngineError):
        QAMModem(4).demodulate_viterbi_hard(np.zeros(10, complex), tr)
    with pytest.raises(_lib.EngineError):
        DeviceGroup()
    with pytest.raises(_lib.EngineError):
        RankComm(0, 1)


def test_ldpc_design_blob_host_only(tmp_path, monkeypatch):
    """design file -> device blob, built on the host (no GPU), deterministic, validated, cached under
    the design file's hash, and identical after a write_ldpc_params -> get_ldpc_code_params round trip
    (ldpc.py:51-141, 257-299)."""
    import ctypes
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import ldpc_params
    from commpy_amd.channelcoding import ldpc as L
    monkeypatch.setenv("CPX_CACHE_DIR", str(tmp_path / "cache"))
    design = os.path.join(ROOT, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt")
    p = L.get_ldpc_code_params(design)
    assert os.listdir(tmp_path / "cache")                                         # compiled design stored under its hash
    q = L.get_ldpc_code_params(design)                                            # second load: from the cache
    gold = ldpc_params("n1944")
    for k in gold:
        assert np.array_equal(np.asarray(p[k]), np.asarray(gold[k])) and np.array_equal(np.asarray(q[k]), np.asarray(gold[k])), k
    blob = p["_cpx_blob"]
    assert blob.dtype == np.uint8 and np.array_equal(blob, q["_cpx_blob"]) and np.array_equal(blob, L.ldpc_design_blob(gold))
    lib = _lib.load()
    nv, nc, ne, mc, mv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
    args = [ctypes.byref(x) for x in (nv, nc, ne, mc, mv)]
    assert lib.cpx_ldpc_blob_info(_lib.ptr(blob), blob.nbytes, *args) == 0
    assert (nv.value, nc.value, ne.value, mc.value, mv.value) == (1944, 648, 7128, 11, 8)
    # the blob holds the (check, variable)-sorted edge list the reference's message_matrix is stored in
    ec, ev = L._edges_from_adjacency(gold)
    words = blob[64:].view(np.int32)
    assert np.array_equal(words[:7128], ec) and np.array_equal(words[7128:2 * 7128], ev)
    # write_ldpc_params -> get_ldpc_code_params round trip gives the same compiled design
    H = np.zeros((648, 1944), np.int8)
    H[ec, ev] = 1
    out = str(tmp_path / "rt.txt")
    L.write_ldpc_params(H, out)
    r = L.get_ldpc_code_params(out)
    assert np.array_equal(r["_cpx_blob"], blob)
    for k in gold:
        assert np.array_equal(np.asarray(r[k]), np.asarray(gold[k])), k
    # validation: truncated, corrupted and tampered blobs are refused (a cache file is untrusted input)
    bad = blob.copy()
    bad[100] ^= 1
    assert lib.cpx_ldpc_blob_info(_lib.ptr(bad), bad.nbytes, *args) == _lib.CPX_EINVAL and "checksum" in _lib.last_error()
    assert lib.cpx_ldpc_blob_info(_lib.ptr(blob), blob.nbytes - 4, *args) == _lib.CPX_EINVAL
    bad = blob.copy()
    bad[0] = ord("X")
    assert lib.cpx_ldpc_blob_info(_lib.ptr(bad), bad.nbytes, *args) == _lib.CPX_EINVAL
    # the reference's shipped designs compile too
    for nm in ("gallager96", "wimax960", "wimax1440"):
        b = L.ldpc_design_blob(ldpc_params(nm))
        assert lib.cpx_ldpc_blob_info(_lib.ptr(b), b.nbytes, *args) == 0 and nv.value == ldpc_params(nm)["n_vnodes"]
    # unsorted / out-of-range edge lists are rejected like cpx_ldpc_create rejects them
    need = ctypes.c_size_t()
    i32p = ctypes.POINTER(ctypes.c_int32)
    e1, e2 = np.array([1, 0], np.int32), np.array([0, 0], np.int32)
    assert lib.cpx_ldpc_blob_build(4, 2, 2, e1.ctypes.data_as(i32p), e2.ctypes.data_as(i32p), None, 0, ctypes.byref(need)) == _lib.CPX_EINVAL
