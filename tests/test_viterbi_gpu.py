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
