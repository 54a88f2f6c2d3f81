"""Per-pair code objects of the codeword-per-lane Viterbi kernels (commpy_amd/jit.py, round 6).

CPU part: hipcc cross-compiles the pair's code object without a GPU; the object must contain exactly the six kernels the library
looks up by name (the mangling pattern of cpx_trellis_attach_viterbi_code / jit.kernel_symbol), the cache must be hit the second
time, and a missing compiler must leave ``None`` + a reason (the table-driven kernel stays).  GPU part: decoded bits with and
without the code object are identical and equal the live reference's goldens; an image of another pair is refused."""
import os
import shutil
import time

import numpy as np
import pytest

from helpers import golden


def test_pair_code_object_compiles_and_holds_the_six_kernels(tmp_path, monkeypatch):
    from commpy_amd import build, jit
    if shutil.which(build._hipcc()) is None and not os.path.exists(build._hipcc()):
        pytest.skip("no hipcc here")
    monkeypatch.setenv("CPX_JIT_CACHE", str(tmp_path))
    lg, g0, g1 = 4, 0o27, 0o35                                    # K = 5: the small-ring flavour, a pair that is not built in
    t0 = time.perf_counter()
    image = jit.viterbi_code_object(lg, g0, g1)
    t_compile = time.perf_counter() - t0
    assert image is not None, jit.viterbi_code_object.last_error
    assert image[:4] == b"\x7fELF"
    for typ in range(3):
        for rt in (False, True):
            assert jit.kernel_symbol(lg, g0, g1, typ, rt).encode() in image
    import re
    pairs = set(re.findall(rb"viterbi_cw_fused_kernelILi(\d+)ELj(\d+)ELj(\d+)E", image))
    assert b"viterbi_cw_acs_kernel" not in image and pairs == {(b"4", b"%d" % g0, b"%d" % g1)}, pairs   # nothing but this pair
    files = os.listdir(tmp_path)
    assert len(files) == 1 and files[0].startswith("vit_") and files[0].endswith("_4_27_35.co"), files
    t0 = time.perf_counter()
    assert jit.viterbi_code_object(lg, g0, g1) == image            # cache hit
    assert time.perf_counter() - t0 < max(0.5, 0.2 * t_compile)
    # no compiler: None and a reason, nothing left in the cache directory
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("CPX_JIT_CACHE", str(tmp_path / "other"))
    monkeypatch.setattr(build, "_hipcc", lambda: "/nonexistent/hipcc")
    assert jit.viterbi_code_object(lg, g0 ^ 2, g1) is None
    assert jit.viterbi_code_object.last_error
    assert os.listdir(tmp_path / "other") == []


@pytest.mark.gpu
def test_pair_code_object_equals_table_driven_kernel_and_reference(gpu, tmp_path, monkeypatch):
    """(135,147) and (165,127), K = 7, and (53,75), K = 6, from tests/golden/viterbi_pairs.npz (decoded by the live reference): the
    forced codeword path runs the table-driven kernel; after Trellis.specialize() the same call runs the pair's code object and
    returns the same bits = the reference's; detach goes back; a code object of ANOTHER pair is refused and changes nothing."""
    import ctypes
    from commpy_amd import _lib, jit
    from commpy_amd.channelcoding import Trellis
    from test_viterbi_cw_gpu import _decode
    g = golden("viterbi_pairs")
    lib = _lib.load()
    images = {}
    for mem, g0, g1 in ((6, 0o135, 0o147), (5, 0o53, 0o75)):
        tr = Trellis(np.array([mem]), np.array([[g0, g1]]))
        base = {}
        for dtype in ("hard", "soft", "unquantized"):
            key = "p%d_%o_%o_%s_None" % (mem, g0, g1, dtype)
            base[dtype] = _decode(g[key + "__rx"], tr, None, dtype, "cw!")
            assert "table-driven" in _lib.last_kernel(), _lib.last_kernel()
            assert np.array_equal(base[dtype], g[key + "__dec"])
        assert tr.specialize() is True, jit.viterbi_code_object.last_error
        h = tr._device_handle()
        assert jit.has_code_object(h)
        for dtype in ("hard", "soft", "unquantized"):
            key = "p%d_%o_%o_%s_None" % (mem, g0, g1, dtype)
            got = _decode(g[key + "__rx"], tr, None, dtype, "cw!")
            assert "code object of this pair" in _lib.last_kernel(), _lib.last_kernel()
            assert np.array_equal(got, base[dtype]) and np.array_equal(got, g[key + "__dec"])
            short = _decode(g[key + "__rx"], tr, 9, dtype, "cw!")             # run-time hop count flavour of the module
            assert "code object of this pair" in _lib.last_kernel() and "runtime hops" in _lib.last_kernel()
            assert np.array_equal(short, _decode(g[key + "__rx"], tr, 9, dtype, "wave"))
        lg, a, b = ctypes.c_int(), ctypes.c_uint(), ctypes.c_uint()
        _lib.check(lib.cpx_trellis_viterbi_spec_query(h, ctypes.byref(lg), ctypes.byref(a), ctypes.byref(b)))
        assert lg.value == 0                                        # nothing more to gain
        _lib.check(lib.cpx_trellis_detach_viterbi_code(h))
        _lib.check(lib.cpx_trellis_viterbi_spec_query(h, ctypes.byref(lg), ctypes.byref(a), ctypes.byref(b)))
        assert lg.value == mem
        images[mem] = jit.viterbi_code_object(lg.value, a.value, b.value)
        _decode(g["p%d_%o_%o_soft_None__rx" % (mem, g0, g1)], tr, None, "soft", "cw!")
        assert "table-driven" in _lib.last_kernel()
    # the K = 6 image on the K = 7 trellis, and the K = 7 image of (135,147) on (165,127): refused, trellis untouched
    tr = Trellis(np.array([6]), np.array([[0o165, 0o127]]))
    h = tr._device_handle()
    for img in (images[5], images[6], b"\x7fELF" + b"\0" * 100):
        buf = ctypes.create_string_buffer(img, len(img))
        with pytest.raises((ValueError, _lib.EngineError)):
            _lib.check(lib.cpx_trellis_attach_viterbi_code(h, buf, len(img)))
        assert not jit.has_code_object(h)
    got = _decode(g["p6_165_127_soft_None__rx"], tr, None, "soft", "cw!")
    assert "table-driven" in _lib.last_kernel() and np.array_equal(got, g["p6_165_127_soft_None__dec"])
    # built-in pairs have nothing to gain
    assert Trellis(np.array([6]), np.array([[0o133, 0o171]])).specialize() is False
