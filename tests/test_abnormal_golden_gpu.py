"""The HIP path against tests/golden/abnormal.npz: what the LIVE reference returns for inputs outside its representable
range (NaN among Viterbi / min-sum LLRs, MAP recursions that underflow, non-finite MAP inputs, turbo in those regimes).
The fast kernels only detect these inputs; flagged codewords / blocks are decoded again by literal kernels (DESIGN.md
"detect and redo").  Integer outputs bit-exact, LLRs: same NaN / +-inf pattern, finite values within 1e-5."""
import numpy as np
import pytest

import oracle

from helpers import Perm, ldpc_params, make_trellis
from test_oracle_golden import abnormal_cases, same_nonfinite_pattern

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_viterbi_nan_inputs(gpu):
    from commpy_amd import _lib
    from commpy_amd.channelcoding import viterbi_decode
    g, names = abnormal_cases("vit_")
    for key in names:
        tname = key[4:key.rindex("_")]
        tr = make_trellis(tname)
        for path in ((None, "cw!", "cw2!", "wave") if tname == "k7_133_171" else (None,)):
            _lib.viterbi_set_path(path)
            try:
                got = viterbi_decode(g[key + "__rx"], tr, None, "soft")
            finally:
                _lib.viterbi_set_path(None)
            assert np.array_equal(got, g[key + "__dec"]), (key, path)


def test_min_sum_nan_llrs(gpu):
    from commpy_amd import _lib
    from commpy_amd.channelcoding import ldpc_bp_decode
    g, names = abnormal_cases("msa_")
    for key in names:
        p = ldpc_params(key[4:])
        for path in ("resident", "tiled"):
            _lib.ldpc_set_path(path)
            try:
                dec, out = ldpc_bp_decode(g[key + "__llr"].copy(), p, "MSA", int(g[key + "__iters"]))
            finally:
                _lib.ldpc_set_path(None)
            assert out.shape == g[key + "__out"].shape
            assert np.array_equal(out, g[key + "__out"], equal_nan=True), (key, path)
            assert np.array_equal(dec, g[key + "__dec"]), (key, path)


def test_map_decode_regimes(gpu):
    from commpy_amd.channelcoding import map_decode
    g, names = abnormal_cases("map_")
    for nm in names:
        key, tname = nm.split("|")
        tr = make_trellis(tname)
        Le, bits = map_decode(g[key + "__sys"], g[key + "__par"], tr, float(g[key + "__nv"]), g[key + "__Lint"], "decode")
        ref, rbits = g[key + "__L"], g[key + "__bits"]
        assert same_nonfinite_pattern(Le, ref), key
        fin = np.isfinite(ref)
        assert np.all(np.abs(Le[fin] - ref[fin]) <= TOL + 1e-9 * np.abs(ref[fin])), key
        assert not np.any((bits != rbits) & ~(np.abs(ref) <= TOL)), key       # NaN / inf positions included


def test_turbo_decode_regimes(gpu):
    from commpy_amd.channelcoding import turbo_decode
    g, names = abnormal_cases("tur_")
    tr = make_trellis("rsc_legacy_4")
    for key in names:
        nv, iters, has_L = g[key + "__par"]
        dec = turbo_decode(g[key + "__sys"], g[key + "__p1"], g[key + "__p2"], tr, float(nv), int(iters),
                           Perm(g[key + "__perm"]), g[key + "__Lint"] if has_L else None)
        assert np.array_equal(dec, g[key + "__dec"]), key


def test_sum_product_zero_llrs(gpu):
    """An LLR of exactly 0 (ldpc.py:214 expects it): the block fills with NaN; dec_word = signbit(out_llrs) and the iteration count
    depend on the NaNs' signs -- generated NaNs negative (x86), tanh / product NaNs positive (NumPy), DESIGN.md 2."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import ldpc_bp_decode
    g, names = abnormal_cases("spaz_")
    for key in names:
        p = ldpc_params(key[5:key.rindex("_")])
        ref = g[key + "__out"]
        fin = np.isfinite(ref)
        for path in ("resident", "tiled"):
            _lib.ldpc_set_path(path)
            try:
                dec, out = ldpc_bp_decode(g[key + "__llr"].copy(), p, "SPA", int(g[key + "__iters"]))
            finally:
                _lib.ldpc_set_path(None)
            assert np.array_equal(np.isnan(out), np.isnan(ref)), (key, path)
            assert np.array_equal(np.signbit(out), np.signbit(ref)), (key, path)
            assert np.all(np.abs(out[fin] - ref[fin]) <= TOL + 1e-6 * np.abs(ref[fin])), (key, path)
            assert np.array_equal(dec, g[key + "__dec"]), (key, path)


def test_flags_resolve_the_codeword_inside_a_full_pair(gpu):
    """Round 4: a flag of the BCJR pass belongs to ONE codeword of the 16 a wave pair decodes (three lane masks: recursion lanes,
    stage / epilogue items).  A batch large enough for full pairs (16 384 codewords) with abnormal codewords planted at every
    position class of a pair -- slot 0, an odd slot, slot 8 (second item set), slot 15, the last codeword of the batch -- and in
    every flag class (a far received pair: stage flag; a contradicting prior of 200: recursion / epilogue flags): the planted
    codewords AND their neighbours equal the oracle (same NaN / inf pattern, 1e-5)."""
    from commpy_amd.channelcoding import map_decode
    tr = make_trellis("rsc_legacy_4")
    rs = np.random.RandomState(99)
    B, N, nv = 16384, 48, 0.1
    sy = rs.randn(B, N) * 0.4 + rs.choice([-1.0, 1.0], size=(B, N))
    pa = rs.randn(B, N) * 0.4 + rs.choice([-1.0, 1.0], size=(B, N))
    li = rs.randn(B, N)
    planted = [0, 3, 16 + 8, 32 + 15, 1000 * 16 + 5, B - 1]
    for i, cw in enumerate(planted):
        if i % 2 == 0:
            sy[cw, 7] = 14.0                                       # worst branch probability below e^-345: flag (A)
            pa[cw, 9] = -17.0
        else:
            li[cw, 11] = 200.0 * np.sign(-sy[cw, 11])              # a prior of e^-200 against the channel: flags (C) - (E)
            sy[cw, 11] *= 6.0
    L, bits = map_decode(sy, pa, tr, nv, li, "decode")
    check = sorted(set(c + d for c in planted for d in (-1, 0, 1) if 0 <= c + d < B))
    for cw in check:
        Lo, bo = oracle.map_decode(sy[cw], pa[cw], tr, nv, li[cw], "decode")
        assert np.array_equal(np.isnan(L[cw]), np.isnan(Lo)) and np.array_equal(np.isposinf(L[cw]), np.isposinf(Lo)) and \
            np.array_equal(np.isneginf(L[cw]), np.isneginf(Lo)), cw
        fin = np.isfinite(Lo)
        assert np.max(np.abs(L[cw][fin] - Lo[fin]), initial=0.0) < 1e-5, (cw, np.max(np.abs(L[cw][fin] - Lo[fin])))
        sure = fin & (np.abs(Lo) > 1e-5)
        assert np.array_equal(bits[cw][sure], bo[sure]), cw


@pytest.mark.parametrize("name", ["two_state", "t57", "rsc_legacy_4", "rsc_matrix_4", "rsc_legacy_8", "k5_23_35"])
def test_literal_wave_kernel_high_snr(gpu, name):
    """Round 5: the redo path of map_decode is the wave-parallel LITERAL kernel (csrc/bcjr.hip map_literal_kernel: the reference's own
    operations on the wave-pair mapping).  sigma^2 = 0.01 flags every codeword (flag (A)); valid codewords give finite LLRs, random
    +-1 symbols (parity contradictions: every step loses e^-100) give the reference's NaN / +-inf pattern.  Every state count the
    fast kernels serve (2, 4 -- shift-register and general --, 8, 16), a block length that is no multiple of the chunk, a batch that
    leaves the last pair partly empty; sampled codewords against the oracle."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import Trellis, conv_encode_batch, map_decode
    tr = Trellis(np.array([1]), np.array([[1, 3]])) if name == "two_state" else make_trellis(name)
    rs = np.random.RandomState(len(name))
    B, N, nv = 16 * 70 + 5, 203, 0.01
    coded = conv_encode_batch(rs.randint(0, 2, (B, N)), tr, "cont")
    sy = 2.0 * coded[:, 0::2] - 1 + np.sqrt(nv) * rs.standard_normal((B, N))
    pa = 2.0 * coded[:, 1::2] - 1 + np.sqrt(nv) * rs.standard_normal((B, N))
    junk = rs.rand(B) < 0.3                                        # not codewords at all
    sy[junk] = rs.choice([-1.0, 1.0], size=(int(junk.sum()), N)) + 0.1 * rs.standard_normal((int(junk.sum()), N))
    li = rs.randn(B, N) * 3.0
    li[rs.rand(B) < 0.1] *= 60.0                                   # priors of e^-+180 and beyond: exp overflow, 1 - p0 cancelling
    L, bits = map_decode(sy, pa, tr, nv, li, "decode")
    note = _lib.last_kernel()
    assert "map_literal_kernel" in note and ("redo: %d of %d" % (B, B)) in note, note
    nan_seen = inf_seen = 0
    for cw in list(range(0, 40)) + list(range(B - 24, B)) + list(rs.randint(0, B, 40)):
        Lo, bo = oracle.map_decode(sy[cw], pa[cw], tr, nv, li[cw], "decode")
        assert np.array_equal(np.isnan(L[cw]), np.isnan(Lo)) and np.array_equal(np.isposinf(L[cw]), np.isposinf(Lo)) and \
            np.array_equal(np.isneginf(L[cw]), np.isneginf(Lo)), (name, cw)
        fin = np.isfinite(Lo)
        assert np.all(np.abs(L[cw][fin] - Lo[fin]) <= TOL + 1e-9 * np.abs(Lo[fin])), (name, cw, np.max(np.abs(L[cw][fin] - Lo[fin])))
        sure = ~fin | (np.abs(Lo) > TOL)
        assert np.array_equal(bits[cw][sure], bo[sure]), (name, cw)
        nan_seen += int(np.isnan(Lo).any())
        inf_seen += int(np.isinf(Lo).any())
    assert nan_seen + inf_seen > 0, "the batch was meant to reach the reference's non-finite regime"


def test_literal_kernel_nothing_flagged_and_long_blocks(gpu):
    """Ordinary inputs raise no flag: the literal launch leaves at once ("redo: 0 of B").  And a block far beyond what the round-3
    per-lane scratch of the redo path admitted still gets its flagged codeword redone (no block-length limit any more)."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import map_decode
    tr = make_trellis("rsc_legacy_4")
    rs = np.random.RandomState(8)
    B, N, nv = 64, 256, 0.6
    sy, pa = rs.randn(B, N) + 1.0, rs.randn(B, N) - 1.0
    L, _ = map_decode(sy, pa, tr, nv, np.zeros((B, N)), "compute")
    assert "redo: 0 of %d" % B in _lib.last_kernel(), _lib.last_kernel()
    assert np.max(np.abs(L[5] - oracle.map_decode(sy[5], pa[5], tr, nv, np.zeros(N), "compute")[0])) < TOL
    B, N = 3, 300000
    sy, pa = rs.randn(B, N) * 0.5 + 1.0, rs.randn(B, N) * 0.5 - 1.0
    sy[1, 1234] = 24.0                                             # flag (A) for codeword 1; its far branches underflow, the near ones do not
    L, _ = map_decode(sy, pa, tr, 0.5, np.zeros((B, N)), "compute")
    assert "redo: 1 of 3" in _lib.last_kernel(), _lib.last_kernel()
    Lo = oracle.map_decode(sy[1], pa[1], tr, 0.5, np.zeros(N), "compute")[0]
    assert np.array_equal(np.isnan(L[1]), np.isnan(Lo)) and np.array_equal(np.isinf(L[1]), np.isinf(Lo))
    fin = np.isfinite(Lo)
    assert fin.sum() > N // 2 and np.max(np.abs(L[1][fin] - Lo[fin])) < TOL
    assert np.max(np.abs(L[2] - oracle.map_decode(sy[2], pa[2], tr, 0.5, np.zeros(N), "compute")[0])) < TOL   # its unflagged pair mates too


@pytest.mark.parametrize("name,n_iter", [("rsc_legacy_4", 3), ("rsc_legacy_8", 2), ("rsc_matrix_4", 1)])
def test_turbo_literal_redo_high_snr(gpu, name, n_iter):
    """Round 5: turbo_decode's redo is ONE launch in which every pair with a flagged codeword runs the reference's whole loop again with
    the literal wave-parallel pass (csrc/bcjr.hip turbo_literal_kernel).  sigma^2 = 0.01 flags everything (flag (A), raised by the slab
    initialisation); a third of the batch is not a codeword at all (NaN / inf LLRs inside the reference's loop, whose comparisons
    `lappr > 0` then read False).  Decoded bits equal the oracle's on sampled codewords, priors included."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import RandInterlv, turbo_decode
    from commpy_amd.devicelink import turbo_encode_gpu
    tr = make_trellis(name)
    rs = np.random.RandomState(40 + n_iter)
    B, N, nv = 16 * 9 + 3, 141, 0.01
    il = RandInterlv(N, 77)
    msgs = rs.randint(0, 2, (B, N))
    s, p1, p2 = (a[:, :N] * 2.0 - 1 + np.sqrt(nv) * rs.standard_normal((B, N)) for a in turbo_encode_gpu(msgs, tr, tr, il))
    junk = rs.rand(B) < 0.3
    s[junk] = rs.choice([-1.0, 1.0], size=(int(junk.sum()), N)) + 0.1 * rs.standard_normal((int(junk.sum()), N))
    Lint = rs.randn(B, N) * 2.0
    for use_L in (False, True):
        dec = turbo_decode(s, p1, p2, tr, nv, n_iter, il, Lint if use_L else None)
        note = _lib.last_kernel()
        assert ("redo: %d of %d" % (B, B)) in note, note
        for cw in list(range(24)) + list(range(B - 8, B)):
            want = oracle.turbo_decode(s[cw], p1[cw], p2[cw], tr, nv, n_iter, il, Lint[cw] if use_L else None)
            assert np.array_equal(dec[cw], want), (name, use_L, cw, int(np.sum(dec[cw] != want)))
    # (what the bits ARE is the reference's business: at this noise variance its probability-domain loop overflows e^L after the first
    # iteration and returns zeros even for clean codewords -- the oracle and the engine agree on that, which is the point)
