"""Randomised parity: random convolutional codes (feed-forward, recursive-systematic, k = 2), random lengths,
traceback depths and metric types -- HIP Viterbi / MAP vs the CPU oracle, bit-exact / 1e-5."""
import warnings

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _random_trellis(rs):
    from commpy_amd.channelcoding import Trellis
    kind = rs.randint(0, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == 0:                                   # feed-forward, k = 1, memory 1..6, n = 2..3
            m = int(rs.randint(1, 7))
            n = int(rs.randint(2, 4))
            g = rs.randint(1, 2 ** (m + 1), (1, n))
            g[0, 0] |= 1 | (1 << m)                     # make the code use its full memory
            return Trellis(np.array([m]), g)
        if kind == 1:                                   # recursive systematic, matrix feedback
            m = int(rs.randint(1, 5))
            fb = int(rs.randint(1, 2 ** (m + 1))) | 1 | (1 << m)
            g = np.array([[1 << 0, int(rs.randint(1, 2 ** (m + 1))) | 1]])
            return Trellis(np.array([m]), g, np.array([[fb]]), 'rsc')
        m1, m2 = int(rs.randint(1, 3)), int(rs.randint(1, 3))   # k = 2, n = 3
        g = rs.randint(0, 2 ** (max(m1, m2) + 1), (2, 3))
        g[0, 0] |= 1
        g[1, 2] |= 1
        return Trellis(np.array([m1, m2]), g)


@pytest.mark.parametrize("seed", range(12))
def test_viterbi_random_codes(gpu, seed):
    from commpy_amd.channelcoding import conv_encode_batch, viterbi_decode
    rs = np.random.RandomState(1000 + seed)
    for _ in range(6):
        tr = _random_trellis(rs)
        if tr.number_states > 128:
            continue
        try:
            tr._device_handle()
        except ValueError:                              # irregular trellis (parallel branches overflow in-degree): reference breaks too
            continue
        nbits = int(rs.randint(8, 200))
        nbits -= nbits % tr.k
        nbits = max(nbits, 4 * tr.k)
        B = int(rs.randint(1, 70))
        term = 'term' if rs.rand() < 0.5 else 'cont'
        coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr, term).astype(float)
        dtype = ('hard', 'soft', 'unquantized')[rs.randint(0, 3)]
        if dtype == 'hard':
            rx = np.where(rs.rand(*coded.shape) < 0.08, 1 - coded, coded)
        elif dtype == 'soft':
            rx = (4.0 * coded - 2) + rs.randn(*coded.shape) * 2.5
        else:
            rx = (2.0 * coded - 1) + rs.randn(*coded.shape)
        L = int(rx.shape[1] * tr.k / tr.n)
        steps = int((L + tr.total_memory) / tr.k) - 1
        tb = None if rs.rand() < 0.4 else int(rs.randint(2, max(3, min(60, steps + 1))))
        got = viterbi_decode(rx, tr, tb, dtype)
        want = oracle.viterbi_decode(rx, tr, tb, dtype)
        assert np.array_equal(got, want), (seed, tr.k, tr.n, tr.number_states, nbits, B, term, dtype, tb)


@pytest.mark.parametrize("seed", range(4))
def test_map_random_rsc_codes(gpu, seed):
    from commpy_amd.channelcoding import Trellis, map_decode
    rs = np.random.RandomState(2000 + seed)
    for _ in range(4):
        m = int(rs.randint(1, 5))
        fb = int(rs.randint(1, 2 ** (m + 1))) | 1 | (1 << m)
        g = np.array([[1, int(rs.randint(1, 2 ** (m + 1))) | 1]])
        tr = Trellis(np.array([m]), g, np.array([[fb]]), 'rsc')
        N, B = int(rs.randint(5, 150)), int(rs.randint(1, 40))
        s, p, li = rs.randn(B, N) * 1.3, rs.randn(B, N) * 1.3, rs.randn(B, N)
        nv = float(rs.uniform(0.3, 2.0))
        L, bits = map_decode(s, p, tr, nv, li, 'decode')
        for b in range(min(B, 3)):
            Lo, bo = oracle.map_decode(s[b], p[b], tr, nv, li[b], 'decode')
            assert np.max(np.abs(L[b] - Lo)) < 1e-5, (seed, m, N)
            assert not np.any((bits[b] != bo) & (np.abs(Lo) > 1e-5))


# ------------------------------------------------------------------ LDPC: degrees beyond the unrolled fast paths
def _random_ldpc(rs, n_v, n_c, row_deg):
    """Random parity-check structure as a reference-style parameter dictionary (ldpc.py:51-141): every check gets
    `row_deg[c]` distinct variables; every variable is used at least once."""
    rows = []
    for c in range(n_c):
        rows.append(np.sort(rs.choice(n_v, size=row_deg[c], replace=False)))
    used = np.zeros(n_v, bool)
    for r in rows:
        used[r] = True
    for v in np.where(~used)[0]:                                   # attach orphans to a random check
        c = rs.randint(n_c)
        rows[c] = np.unique(np.append(rows[c], v))
    cdeg = np.array([len(r) for r in rows], np.int32)
    cols = [[] for _ in range(n_v)]
    for c, r in enumerate(rows):
        for v in r:
            cols[v].append(c)
    vdeg = np.array([len(x) for x in cols], np.int32)
    mcd, mvd = int(cdeg.max()), int(vdeg.max())
    cadj = -np.ones((n_c, mcd), int)
    vadj = -np.ones((n_v, mvd), int)
    for c, r in enumerate(rows):
        cadj[c, :len(r)] = r
    for v, x in enumerate(cols):
        vadj[v, :len(x)] = x
    cvm = -np.ones((n_c, mcd), int)
    vcm = -np.ones((n_v, mvd), int)
    for c in range(n_c):
        for i, v in enumerate(cadj[c, :cdeg[c]]):
            cvm[c, i] = np.where(vadj[v] == c)[0][0]
    for v in range(n_v):
        for i, c in enumerate(vadj[v, :vdeg[v]]):
            vcm[v, i] = np.where(cadj[c] == v)[0][0]
    import scipy.sparse as sp
    H = sp.lil_matrix((n_c, n_v), dtype=np.int8)
    for c, r in enumerate(rows):
        H[c, r] = 1
    # parity_check_matrix given up front: the reference only builds it (with an LU factorisation that needs a
    # structured code) when the key is missing (ldpc.py:189-195)
    return {"parity_check_matrix": H.tocsc(),
            "n_vnodes": n_v, "n_cnodes": n_c, "max_cnode_deg": mcd, "max_vnode_deg": mvd,
            "cnode_adj_list": cadj.flatten().astype(np.int32), "cnode_vnode_map": cvm.flatten().astype(np.int32),
            "vnode_adj_list": vadj.flatten().astype(np.int32), "vnode_cnode_map": vcm.flatten().astype(np.int32),
            "cnode_deg_list": cdeg, "vnode_deg_list": vdeg}


@pytest.mark.parametrize("seed,n_v,n_c,lo,hi", [(1, 120, 40, 2, 7), (2, 200, 90, 9, 20), (3, 96, 64, 13, 31),
                                                (4, 200, 24, 24, 28)])
def test_ldpc_random_structures(gpu, seed, n_v, n_c, lo, hi):
    """Check degrees 2..32 (unrolled exact-degree paths up to 12, the rolled path above, MAXDEG = 32 itself) and
    variable degrees beyond one four-edge chunk; ragged batch; both algorithms against the oracle."""
    from commpy_amd.channelcoding import ldpc_bp_decode
    from oracle import ldpc_bp_decode as ref_decode
    rs = np.random.RandomState(500 + seed)
    p = _random_ldpc(rs, n_v, n_c, rs.randint(lo, hi + 1, size=n_c))
    assert int(p["max_vnode_deg"]) > 4 or seed == 1
    B = 131
    llr = (rs.randn(B * n_v) * 3.0 + 1.5)
    llr[rs.randint(0, llr.size, 40)] = 0.0                         # exact zeros: sign(0) = 0 / min = 0 paths
    for alg, iters in (("MSA", 7), ("SPA", 4)):
        dec, out, its = ldpc_bp_decode(llr.copy(), dict(p), alg, iters, return_iterations=True)
        do, oo, io = ref_decode(llr.copy(), dict(p), alg, iters, True)
        assert np.array_equal(its, io), alg
        if alg == "MSA":
            assert np.array_equal(out, oo)
            assert np.array_equal(dec, do)
        else:
            fin = np.isfinite(oo)
            assert np.array_equal(np.isfinite(out), fin)
            close = np.abs(out[fin] - oo[fin]) <= 1e-5 + 1e-6 * np.abs(oo[fin])
            assert np.mean(close) > 0.999, np.max(np.abs(out[fin] - oo[fin]))


@pytest.mark.parametrize("seed,n_v,n_c,lo,hi", [(3, 96, 64, 13, 31), (4, 200, 24, 24, 28), (6, 150, 60, 2, 32)])
def test_ldpc_rolled_rows_on_the_tiled_path(gpu, seed, n_v, n_c, lo, hi):
    """Forced onto the tiled (beyond-LDS) kernels, rows of 13 .. 32 edges take the rolled form of both check passes -- eight edges'
    operands requested at a time since round 5, the last chunk ragged.  Same operations in the same order as the LDS-resident
    log-domain row: identical iteration counts, dec_word and out_llrs; and against the oracle."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import ldpc_bp_decode
    from oracle import ldpc_bp_decode as ref_decode
    rs = np.random.RandomState(500 + seed)
    p = _random_ldpc(rs, n_v, n_c, rs.randint(lo, hi + 1, size=n_c))
    B = 131
    llr = (rs.randn(B * n_v) * 3.0 + 1.5)
    llr[rs.randint(0, llr.size, 40)] = 0.0
    for alg, iters in (("MSA", 7), ("SPA", 4)):
        got = {}
        for path in ("tiled", "resident-log"):
            _lib.ldpc_set_path(path)
            try:
                got[path] = ldpc_bp_decode(llr.copy(), dict(p), alg, iters, return_iterations=True)
                assert ("tiled" in _lib.last_kernel()) == (path == "tiled"), _lib.last_kernel()
            finally:
                _lib.ldpc_set_path(None)
        for a, b in zip(got["tiled"], got["resident-log"]):
            assert np.array_equal(a, b, equal_nan=True), alg
        dec, out, its = got["tiled"]
        do, oo, io = ref_decode(llr.copy(), dict(p), alg, iters, True)
        assert np.array_equal(its, io), alg
        if alg == "MSA":
            assert np.array_equal(out, oo) and np.array_equal(dec, do)
        else:
            fin = np.isfinite(oo)
            assert np.array_equal(np.isfinite(out), fin)
            assert np.mean(np.abs(out[fin] - oo[fin]) <= 1e-5 + 1e-6 * np.abs(oo[fin])) > 0.999


def test_ldpc_check_degree_above_32_takes_the_general_kernel(gpu):
    """Round 4: a check of more than 32 edges (rows are kept in registers / a 32-bit sign mask by the fast kernels) no longer
    raises: the literal kernel decodes the whole code, min-sum bit-identical to the oracle, sum-product within 1e-5."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import ldpc_bp_decode
    rs = np.random.RandomState(9)
    p = _random_ldpc(rs, 80, 4, np.array([40, 33, 10, 12]))
    for alg in ("MSA", "SPA"):
        llr = rs.randn(80 * 3) * 2.0 + 1.5
        dec, out, its = ldpc_bp_decode(llr.copy(), p, alg, 6, return_iterations=True)
        assert "ldpc_exact_kernel" in _lib.last_kernel()
        do, oo, io = oracle.ldpc_bp_decode(llr.copy(), p, alg, 6, return_iters=True)
        assert np.array_equal(its, io) and np.array_equal(dec, do)
        assert np.array_equal(out, oo) if alg == "MSA" else np.max(np.abs(out - oo)) <= 1e-5
