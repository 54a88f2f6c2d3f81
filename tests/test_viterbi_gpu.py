"""GPU parity of the HIP Viterbi decoder (through the C-ABI) against the golden vectors generated
from the live reference and against the CPU oracle.  Bit-exact for every decoding type."""
import numpy as np
import pytest

import oracle
from helpers import TableTrellis, golden, make_trellis

pytestmark = pytest.mark.gpu


def _decode(x, tr, tb, dtype):
    from commpy_amd.channelcoding import viterbi_decode
    return viterbi_decode(x, tr, tb, dtype)


def test_golden_grid_bit_exact(gpu):
    """576 reference cases: 12 trellises x hard/soft/unquantized x term/cont x tb x noise (+-inf LLRs included)."""
    g = golden("viterbi_small")
    bad = []
    done = 0
    for nm in g["names"]:
        key, tname, term, dtype, tb, noisy = str(nm).split("|")
        tr = make_trellis(tname)
        tb = None if tb == "None" else int(tb)
        dec = _decode(g[key + "__in"], tr, tb, dtype)
        done += 1
        if dec.dtype != np.int64 or not np.array_equal(dec, g[key + "__out"]):
            bad.append(str(nm))
    assert done == len(g["names"]) >= 500
    assert not bad, "mismatching cases: %s" % bad[:10]


def test_config1_hard_bsc_batch(gpu):
    """BASELINE config 1 (K=3 [[5,7]], 64-bit blocks, hard/BSC) as ONE batched call."""
    c1 = golden("viterbi_c1")
    dec = _decode(c1["rx"], make_trellis("t57"), None, "hard")
    assert dec.shape == c1["dec"].shape
    assert np.array_equal(dec, c1["dec"])


def test_config2_soft_k7_reference_vectors(gpu):
    """BASELINE config 2 (K=7 (133,171), 1024-bit, soft LLRs from QPSK+AWGN): reference outputs."""
    c2 = golden("viterbi_c2")
    tr = make_trellis("k7_133_171")
    for tag in ("e3", "e1"):
        dec = _decode(c2[tag + "__llr"], tr, None, "soft")
        assert np.array_equal(dec, c2[tag + "__dec"]), tag


@pytest.mark.parametrize("tname,dtype", [("k7_133_171", "soft"), ("k7_133_171", "hard"), ("k5_23_35", "unquantized"),
                                         ("k2_default", "soft"), ("rsc_legacy_8", "soft"), ("t57", "hard"),
                                         ("k2_rsc_matrix", "hard"), ("r13_k4", "soft"), ("k8_247_371", "soft"),
                                         ("k8_247_371", "hard")])
def test_random_batches_vs_oracle(gpu, tname, dtype):
    """Seeded random batches (ragged last wave, several tb depths) against the CPU oracle, bit-exact."""
    from commpy_amd.channelcoding import conv_encode
    tr = make_trellis(tname)
    rs = np.random.RandomState(hash((tname, dtype)) % (2 ** 31))
    for B, nbits, tb in ((37, 120, None), (5, 333, 15), (130, 96, 40)):
        nbits -= nbits % tr.k
        msgs = rs.randint(0, 2, (B, nbits))
        coded = np.stack([conv_encode(m, tr) for m in msgs]).astype(float)
        if dtype == "hard":
            rx = np.where(rs.rand(*coded.shape) < 0.05, 1 - coded, coded)
        elif dtype == "soft":
            rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 2.0
        else:
            rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.8
        got = _decode(rx, tr, tb, dtype)
        want = oracle.viterbi_decode(rx, tr, tb, dtype)
        assert np.array_equal(got, want), (tname, dtype, B, nbits, tb)


def test_full_size_roundtrip_properties(gpu):
    """BASELINE config-2 size (B=65536 x 1024 bits): noiseless encode -> decode returns the message
    exactly (size-independent property), and a noisy slice matches the oracle."""
    from commpy_amd.channelcoding import conv_encode
    tr = make_trellis("k7_133_171")
    rs = np.random.RandomState(10)
    base = rs.randint(0, 2, (64, 1024))
    coded = np.stack([conv_encode(m, tr) for m in base]).astype(float)
    B = 65536
    msgs = np.tile(base, (B // 64, 1))
    llr = np.tile(8.0 * coded - 4.0, (B // 64, 1))
    dec = _decode(llr, tr, None, "soft")
    assert dec.shape == (B, 1030)
    assert np.array_equal(dec[:, :1024], msgs)
    assert not dec[:, 1024:].any()
    noisy = llr[:256] + rs.randn(256, llr.shape[1]) * 3.0
    assert np.array_equal(_decode(noisy, tr, None, "soft"), oracle.viterbi_decode(noisy, tr, None, "soft"))


def test_invalid_arguments(gpu):
    tr = make_trellis("t57")
    with pytest.raises(ValueError):
        _decode(np.zeros(20), tr, None, "fuzzy")
    out = _decode(np.zeros((0, 20)), tr, None, "hard")
    assert out.shape == (0, 10)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["k7_133_171", "t57", "k2_default", "k8_247_371"])
def test_soft_nan_inputs_follow_the_reference(gpu, name):
    """The reference's clip lets a NaN LLR through (convcode.py:719): from that step on every metric of the codeword is NaN,
    every decision "first predecessor", every traceback starts from state 0 (:633-645).  The fast kernels only detect the NaN;
    flagged codewords are decoded again by the NaN-exact instantiation (csrc/viterbi.hip, "NaN among 'soft' inputs").  All
    kernel paths (fused / two-kernel codeword path, state-per-lane, 128-state) against the oracle, which was checked against
    the live reference on such inputs."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import viterbi_decode
    tr = make_trellis(name)
    rs = np.random.RandomState(5)
    for B, steps in ((1, 40), (70, 130), (64, 64), (200, 33)):
        rx = rs.randn(B, steps * tr.n) * 3
        rx[rs.rand(*rx.shape) < 0.004] = np.nan
        rx[0, rs.randint(0, rx.shape[1])] = np.nan
        if B > 100:
            rx[100:] = rs.randn(B - 100, steps * tr.n) * 3          # whole wavefronts / redo groups without a NaN
        want = oracle.viterbi_decode(rx, tr, None, "soft")
        for path in ((None, "cw!", "cw2!", "wave") if name == "k7_133_171" else (None,)):
            _lib.viterbi_set_path(path)
            try:
                got = viterbi_decode(rx, tr, None, "soft")
            finally:
                _lib.viterbi_set_path(None)
            assert np.array_equal(got, want), (name, B, steps, path)


@pytest.mark.gpu
def test_abnormal_inputs_all_types_vs_oracle(gpu):
    """Inputs nobody should send (promoted from scripts/micro/vit_weird.py): non-binary 'hard' values, +-inf / NaN / 1e200 in
    'unquantized', +-inf / NaN / +-500 / +-0 in 'soft'; five trellises (k = 1 and 2, 4 to 128 states), every kernel path."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import viterbi_decode
    rs = np.random.RandomState(0)
    n_cases = 0
    for name in ("k7_133_171", "t57", "k2_default", "rsc_legacy_8", "k8_247_371"):
        tr = make_trellis(name)
        for trial in range(12):
            B, steps = int(rs.choice([1, 5, 64, 70])), int(rs.randint(20, 150))
            length = steps * tr.n
            for dtype in ("hard", "unquantized", "soft"):
                if dtype == "hard":
                    rx = rs.choice([0.0, 1.0, 2.0, -1.0, 0.5, 1.9, 3.0, -0.3], size=(B, length), p=[.4, .4, .04, .04, .03, .03, .03, .03])
                elif dtype == "unquantized":
                    rx = rs.choice([-1.0, 1.0], size=(B, length)) + rs.randn(B, length) * 0.6
                    for v in (np.inf, -np.inf, np.nan, 1e200, -1e200, 0.0):
                        rx[rs.rand(B, length) < 0.004] = v
                else:
                    rx = rs.randn(B, length) * 4
                    for v in (np.inf, -np.inf, np.nan, 1e200, 499.99999, -500.0, 0.0, -0.0):
                        rx[rs.rand(B, length) < 0.004] = v
                want = oracle.viterbi_decode(rx, tr, None, dtype)
                for path in ((None, "cw!", "cw2!", "wave") if name == "k7_133_171" else (None,)):
                    _lib.viterbi_set_path(path)
                    try:
                        got = viterbi_decode(rx, tr, None, dtype)
                    finally:
                        _lib.viterbi_set_path(None)
                    n_cases += 1
                    assert np.array_equal(got, want), (name, dtype, B, steps, path, int(np.sum(got != want)))
    assert n_cases >= 250
