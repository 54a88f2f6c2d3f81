"""LDS-resident LDPC path (csrc/ldpc_resident.hip) against the tiled HBM path (csrc/ldpc.hip) and the oracle.

Both paths replace ldpc_bp_decode (/root/reference/commpy/channelcoding/ldpc.py:144-254).  Min-sum, and sum-product with the
log-domain row ('tiled' and 'resident-log'): the same float64 operations per edge in the same order, so dec_word, out_llrs and
the executed iterations must be IDENTICAL -- whatever slot of whatever workgroup a block lands in.  Sum-product on the default
resident path (round 4: ldpc_resident_ratio_kernel, the state kept as likelihood ratios, no exp / log inside an iteration): the
same dec_word, iteration counts and NaN pattern, out_llrs inside the banded contract of helpers.spa_contract against the
log-domain row -- its arithmetic differs by ~1e-16 relative per operation, which 2 atanh amplifies near saturation exactly as
for the reference's own row.  The other LDPC tests of the suite run through the default path and compare with the oracle;
here the paths are forced and compared with each other on batches that exercise slot replacement (more blocks than slots,
blocks that converge at very different iterations, blocks that never converge), plus special values.
"""
import numpy as np
import pytest

import oracle
from helpers import ldpc_params

pytestmark = pytest.mark.gpu


@pytest.fixture
def paths():
    from commpy_amd import _lib
    yield _lib
    _lib.ldpc_set_path(None)


def _decode(_lib, path, llr, p, alg, iters):
    from commpy_amd.channelcoding import ldpc_bp_decode
    _lib.ldpc_set_path(path)
    x = llr.copy()
    dec, out, its = ldpc_bp_decode(x, p, alg, iters, return_iterations=True)
    return dec, out, its, x, _lib.last_kernel()


def _ratio_agrees(a, b, what):
    """Sum-product, ratio-domain kernel `a` against the log-domain row `b`: (dec, out, iters) each."""
    from helpers import spa_contract, spa_rows_agree
    (d1, o1, i1), (d2, o2, i2) = a, b
    assert np.array_equal(i1, i2), what
    assert np.array_equal(np.isnan(o1), np.isnan(o2)), what
    ok = ~np.isnan(o2)                                             # the sign bit of a NaN is nobody's contract (np.signbit of it, :248)
    assert np.array_equal(d1[ok], d2[ok]), what
    assert np.array_equal(np.isinf(o1), np.isinf(o2)) and np.array_equal(o1[np.isinf(o2)], o2[np.isinf(o2)]), what
    fin = np.isfinite(o2)
    spa_contract(o1[fin], o2[fin], what)
    spa_rows_agree(o1[fin], o2[fin], what)                         # round 5: the tight row-vs-row bound (helpers.SPA_ROW_BANDS)


def _staggered(rs, B, n, rate, ebn0s):
    ebn0 = rs.choice(ebn0s, size=B)
    sig = 1 / np.sqrt(10 ** (ebn0 / 10.0) * rate * 2)
    return (2.0 * (1.0 + sig[:, None] * rs.randn(B, n)) / sig[:, None] ** 2).reshape(-1)


@pytest.mark.parametrize("alg,iters", [("MSA", 50), ("SPA", 14), ("MSA", 1), ("SPA", 2)])
def test_resident_equals_tiled_1944(gpu, paths, alg, iters):
    p = ldpc_params("n1944")
    rs = np.random.RandomState(5)
    B = 2100                                                        # > 256 CUs x 4 slots: every workgroup refills slots
    llr = _staggered(rs, B, 1944, 2.0 / 3, [0.5, 2.0, 2.6, 3.2, 4.5, 30.0])
    d1, o1, i1, x1, k1 = _decode(paths, "tiled", llr, p, alg, iters)
    d2, o2, i2, x2, k2 = _decode(paths, "resident", llr, p, alg, iters)
    assert "tiled" in k1 and "ldpc_resident_kernel" in k2 and alg in k2, (k1, k2)
    assert len(set(i1.tolist())) >= min(iters, 5)
    assert np.array_equal(x1, x2)
    if alg == "SPA":
        assert "ratio" in k2, k2
        _ratio_agrees((d2, o2, i2), (d1, o1, i1), "ratio kernel vs tiled")
        d2, o2, i2, x2, k2 = _decode(paths, "resident-log", llr, p, alg, iters)
        assert "ldpc_resident_kernel<SPA,0>" in k2, k2
    assert np.array_equal(i1, i2)
    assert np.array_equal(d1, d2)
    assert np.array_equal(o1, o2)                                   # bit-identical, both algorithms
    assert np.array_equal(x1, x2)


@pytest.mark.parametrize("name,n", [("gallager96", 96), ("wimax1440", 1440)])
def test_resident_other_codes_vs_oracle_and_tiled(gpu, paths, name, n):
    """Smaller codes take more slots per workgroup (G = 16 / 4)."""
    p = ldpc_params(name)
    rs = np.random.RandomState(11)
    B = 777
    llr = _staggered(rs, B, n, 0.5, [1.0, 2.5, 4.0, 30.0])
    for alg, iters in (("MSA", 20), ("SPA", 8)):
        d1, o1, i1, _, _ = _decode(paths, "tiled", llr, p, alg, iters)
        d2, o2, i2, _, k2 = _decode(paths, "resident", llr, p, alg, iters)
        assert "ldpc_resident_kernel" in k2
        if alg == "SPA":
            _ratio_agrees((d2, o2, i2), (d1, o1, i1), name)
            d3, o3, i3, _, _ = _decode(paths, "resident-log", llr, p, alg, iters)
            assert np.array_equal(i1, i3) and np.array_equal(d1, d3) and np.array_equal(o1, o3), (name, alg)
        else:
            assert np.array_equal(i1, i2) and np.array_equal(d1, d2) and np.array_equal(o1, o2), (name, alg)
        sel = slice(0, 64 * n)
        do, oo, io = oracle.ldpc_bp_decode(llr[sel].copy(), p, alg, iters, True)
        assert np.array_equal(i2[:64], io) and np.array_equal(d2[:, :64], do), (name, alg)
        if alg == "MSA":
            assert np.array_equal(o2[:, :64], oo)


@pytest.mark.parametrize("B", [1, 3, 5, 64, 65])
def test_resident_small_batches_and_special_values(gpu, paths, B):
    """Fewer blocks than slots; +-inf, NaN, zeros, values beyond the +-500 clip (in-place clip, ldpc.py:186)."""
    p = ldpc_params("n1944")
    rs = np.random.RandomState(100 + B)
    llr = _staggered(rs, B, 1944, 2.0 / 3, [2.0, 3.5, 30.0])
    llr[rs.randint(llr.size, size=40)] = 0.0
    llr[rs.randint(llr.size, size=10)] = -0.0
    llr[rs.randint(llr.size, size=10)] = 1e4
    llr[rs.randint(llr.size, size=10)] = -np.inf
    if B >= 3:
        llr[1944 * 2 + 7] = np.nan
    for alg, iters in (("MSA", 9), ("SPA", 5)):
        d1, o1, i1, x1, _ = _decode(paths, "tiled", llr, p, alg, iters)
        d2, o2, i2, x2, _ = _decode(paths, "resident", llr, p, alg, iters)
        if alg == "SPA":                                           # blocks with a zero LLR or a NaN come back through the log-domain kernel
            _ratio_agrees((d2, o2, i2), (d1, o1, i1), "special values")
            assert np.array_equal(x1, x2, equal_nan=True)
            d2, o2, i2, x2, _ = _decode(paths, "resident-log", llr, p, alg, iters)
        assert np.array_equal(i1, i2), alg
        assert np.array_equal(o1, o2, equal_nan=True), alg
        ok = ~np.isnan(o1)                                         # the sign bit of a NaN is nobody's contract (np.signbit of it, :248)
        assert np.array_equal(d1[ok], d2[ok]), alg
        assert np.array_equal(x1, x2, equal_nan=True) and np.nanmax(np.abs(x2)) <= 500.0


def test_resident_strict_mode_and_zero_iterations(gpu, paths):
    from commpy_amd.channelcoding import ldpc_bp_decode
    p = ldpc_params("gallager96")
    llr = np.random.RandomState(3).randn(96 * 4) * 3
    paths.ldpc_set_path("resident")
    with pytest.raises(Exception):
        ldpc_bp_decode(llr.copy(), p, "MSA", 0)                     # n_iters == 0 is the tiled path's job; strict mode says so
    paths.ldpc_set_path(None)
    dec, out = ldpc_bp_decode(llr.copy(), p, "MSA", 0)
    assert np.array_equal(out.reshape(-1, order="F"), llr)          # out_llrs = llr (ldpc.py:194)
    with pytest.raises(Exception):
        paths.ldpc_set_path("bogus")


def test_auto_falls_back_to_tiled_for_codes_beyond_lds(gpu, paths):
    """A Tanner graph whose per-block state (Q + one message per edge) exceeds the 160 KB of a compute unit: 'auto' takes the
    tiled HBM path, 'resident' refuses, results against the oracle."""
    from test_random_codes_gpu import _random_ldpc
    from commpy_amd.channelcoding import ldpc_bp_decode
    rs = np.random.RandomState(8)
    n_v, n_c = 9000, 4500
    p = _random_ldpc(rs, n_v, n_c, rs.randint(3, 6, size=n_c))      # ~18 000 edges + 9 000 LLRs > 160 KB
    B = 5
    llr = (2.0 + 1.3 * rs.randn(B * n_v)) * 1.5
    for alg, iters in (("MSA", 6), ("SPA", 4)):
        dec, out, its = ldpc_bp_decode(llr.copy(), dict(p), alg, iters, return_iterations=True)
        assert "tiled" in paths.last_kernel()
        do, oo, io = oracle.ldpc_bp_decode(llr.copy(), dict(p), alg, iters, True)
        assert np.array_equal(its, io) and np.array_equal(dec, do), alg
        if alg == "MSA":
            assert np.array_equal(out, oo)
        else:
            assert np.all(np.abs(out - oo) <= 1e-5 + 1e-6 * np.abs(oo))
    paths.ldpc_set_path("resident")
    with pytest.raises(Exception):
        ldpc_bp_decode(llr.copy(), dict(p), "MSA", 3)


@pytest.mark.parametrize("name,n,iters", [("n1944", 1944, 12), ("gallager96", 96, 30), ("wimax1440", 1440, 8)])
def test_min_sum_nan_llrs_follow_the_reference(gpu, paths, name, n, iters):
    """NumPy's min / sign propagate a NaN (ldpc.py:229-238): within a few iterations the whole block is NaN and dec_word =
    signbit(NaN) decides the syndrome test.  The fast kernels only detect a NaN LLR; flagged blocks are decoded again by the
    literal ldpc_msa_exact_kernel (csrc/ldpc.hip).  Both paths against the oracle: iteration counts, dec_word, out_llrs and its
    NaN pattern -- np.nan inputs (one NaN sign; see the kernel's note on mixed signs)."""
    p = ldpc_params(name)
    rs = np.random.RandomState(21)
    B = 70
    llr = _staggered(rs, B, n, 0.5, [1.0, 2.5, 4.0, 30.0])
    blocks = [0, 3, 5, 64, 69]
    for b in blocks:
        llr[b * n + rs.randint(n, size=1 + (b % 3))] = np.nan
    llr[5 * n:6 * n] = np.nan                                       # a block of nothing but NaN
    do, oo, io = oracle.ldpc_bp_decode(llr.copy(), p, "MSA", iters, True)
    assert np.isnan(oo[:, blocks]).any(axis=0).all() and not np.isnan(np.delete(oo, blocks, axis=1)).any()
    for path in ("resident", "tiled"):
        d, o, i, x, _ = _decode(paths, path, llr, p, "MSA", iters)
        assert np.array_equal(i, io), path
        assert np.array_equal(o, oo, equal_nan=True), path
        assert np.array_equal(d, do), path
        assert np.array_equal(x, np.clip(llr, -500, 500), equal_nan=True), path      # in-place clip (:186); np.clip lets NaN through


@pytest.mark.parametrize("path", ["resident", "tiled"])
def test_block_major_outputs_equal_the_column_layout(gpu, paths, path):
    """cpx_ldpc_bp_decode_batch_bm[_dev] (dec_word / out_llrs [B][n_v], one block per row -- the memory behind the reference's
    F-ordered result, ldpc.py:251-253) against the [n_v][B] entry point: identical values and iteration counts for both
    algorithms, NaN blocks (min-sum's exact redo kernel writes through the same strides) and a ragged batch; and the Python
    function returns arrays with the reference's shape AND strides."""
    import ctypes
    from commpy_amd.channelcoding import ldpc_bp_decode
    from commpy_amd.channelcoding.ldpc import _device_code
    from commpy_amd.devicelink import DeviceBuf
    lib = paths.load()
    p = ldpc_params("wimax1440")
    n, B = 1440, 131
    rs = np.random.RandomState(17)
    llr = _staggered(rs, B, n, 0.5, [1.0, 2.5, 4.0, 30.0])
    llr[7 * n + 11] = np.nan
    llr[130 * n + 5] = np.nan
    paths.ldpc_set_path(path)
    code = _device_code(p)
    for alg, name, iters in ((1, "MSA", 12), (0, "SPA", 6)):
        res = []
        for fn in (lib.cpx_ldpc_bp_decode_batch_dev, lib.cpx_ldpc_bp_decode_batch_bm_dev):
            d_llr = DeviceBuf.from_array(llr)
            d_dec, d_out, d_it = DeviceBuf(B * n), DeviceBuf(B * n * 8), DeviceBuf(B * 4)
            paths.check(fn(code, d_llr.ptr, B, alg, iters, d_dec.ptr, d_out.ptr, d_it.ptr, None))
            paths.check(lib.cpx_stream_sync(None))
            res.append((d_dec, d_out, d_it))
        dec_c, out_c = res[0][0].to_array((n, B), np.int8), res[0][1].to_array((n, B), np.float64)
        dec_b, out_b = res[1][0].to_array((B, n), np.int8), res[1][1].to_array((B, n), np.float64)
        assert np.array_equal(res[0][2].to_array((B,), np.int32), res[1][2].to_array((B,), np.int32)), name
        assert np.array_equal(dec_b.T, dec_c) and np.array_equal(out_b.T, out_c, equal_nan=True), name
        for r in res:
            for d in r:
                d.free()
        dec, out = ldpc_bp_decode(llr.copy(), p, name, iters)
        assert dec.shape == (n, B) and out.shape == (n, B) and dec.dtype == np.int8 and out.dtype == np.float64
        assert dec.flags.f_contiguous and out.flags.f_contiguous and out.strides == (8, 8 * n)     # the reference's reshape(order='F')
        assert np.array_equal(dec, dec_c) and np.array_equal(out, out_c, equal_nan=True), name
