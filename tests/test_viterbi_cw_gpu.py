"""The codeword-per-lane Viterbi path (csrc/viterbi_cw.hip) against the reference goldens, the CPU oracle and the
state-per-lane kernels -- bit-exact for every decoding type.  cpx_viterbi_set_path forces a path: "cw!" = codeword path or
fail (the fused single kernel for tb_depth <= 48, else ACS + traceback kernels), "cw2!" = always the two-kernel form,
"wave" = state-per-lane kernels."""
import os

import numpy as np
import pytest

import oracle
from helpers import golden, make_trellis

pytestmark = pytest.mark.gpu


class _path:
    """Force a Viterbi kernel path for the duration of a ``with`` block (cpx_viterbi_set_path)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        from commpy_amd import _lib
        _lib.viterbi_set_path(self.name)

    def __exit__(self, *a):
        from commpy_amd import _lib
        _lib.viterbi_set_path(None)


def _decode(x, tr, tb, dtype, path):
    from commpy_amd.channelcoding import viterbi_decode
    with _path(path):
        return viterbi_decode(x, tr, tb, dtype)


def test_forced_path_is_really_taken(gpu):
    """"cw!" must refuse a trellis it has no instantiation for instead of silently using the wave kernels."""
    tr = make_trellis("k8_247_371")
    with pytest.raises(ValueError):
        _decode(np.zeros((2, 60)), tr, None, "hard", "cw!")
    with pytest.raises(ValueError):
        _decode(np.zeros((2, 60)), make_trellis("rsc_legacy_4"), None, "hard", "cw!")   # recursive: not a shift register
    tr = make_trellis("k7_133_171")
    with pytest.raises(ValueError):
        _decode(np.zeros((2, 61)), tr, None, "hard", "cw!")          # odd row length: rows not 16-byte aligned
    assert _decode(np.zeros((2, 60)), tr, None, "hard", "cw!").shape == (2, 30)


CW_TRELLISES = ("k7_133_171",)     # the instantiated code: K = 7 (133,171)


def test_golden_cases_through_the_codeword_path(gpu):
    """Every case of the reference grid for the instantiated codes (hard/soft/unquantized x term/cont x tb x noise,
    +-inf LLRs)."""
    g = golden("viterbi_small")
    done, bad = 0, []
    for nm in g["names"]:
        key, tname, term, dtype, tb, noisy = str(nm).split("|")
        if tname not in CW_TRELLISES or len(g[key + "__in"]) % 2:
            continue
        tr = make_trellis(tname)
        tb = None if tb == "None" else int(tb)
        done += 1
        for path in ("cw!", "cw2!"):
            dec = _decode(g[key + "__in"], tr, tb, dtype, path)
            if dec.dtype != np.int64 or not np.array_equal(dec, g[key + "__out"]):
                bad.append(str(nm) + path)
    assert done >= 24, done
    assert not bad, bad[:10]
    c2 = golden("viterbi_c2")
    tr = make_trellis("k7_133_171")
    for tag in ("e3", "e1"):
        for path in ("cw!", "cw2!"):
            assert np.array_equal(_decode(c2[tag + "__llr"], tr, None, "soft", path), c2[tag + "__dec"]), (tag, path)


@pytest.mark.parametrize("dtype", ["hard", "soft", "unquantized"])
@pytest.mark.parametrize("B,nbits,tb", [(1, 1, None), (3, 2, 2), (63, 5, None), (64, 30, None), (65, 31, 15), (130, 96, 40),
                                        (37, 120, 48), (200, 64, 3), (5, 333, None), (70, 1024, None), (257, 24, 30),
                                        (100, 25, 30), (300, 95, None), (64, 96, 30), (129, 97, 30), (9, 191, 30)])
def test_random_batches_vs_oracle(gpu, dtype, B, nbits, tb):
    """Seeded batches, ragged groups, window edges (tb = 2, 3, window = 64 KiB limit), ties (hard) -- vs the CPU oracle."""
    from commpy_amd.channelcoding import conv_encode_batch
    tr = make_trellis("k7_133_171")
    rs = np.random.RandomState(1000 * B + nbits)
    coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
    if dtype == "hard":
        rx = np.where(rs.rand(*coded.shape) < 0.08, 1 - coded, coded)
    elif dtype == "soft":
        rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 2.0
        rx[rs.rand(*rx.shape) < 0.01] = np.inf
        rx[rs.rand(*rx.shape) < 0.01] = -np.inf
        rx[rs.rand(*rx.shape) < 0.01] = 0.0
    else:
        rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.8
    want = oracle.viterbi_decode(rx, tr, tb, dtype)
    for path in ("cw!", "cw2!", "wave"):
        assert np.array_equal(_decode(rx, tr, tb, dtype, path), want), (dtype, B, nbits, tb, path)


@pytest.mark.parametrize("dtype", ["hard", "soft", "unquantized"])
@pytest.mark.parametrize("tb", [31, 36, 47, 48, 49])
def test_deep_traceback_stays_in_the_fused_kernel(gpu, dtype, tb):
    """tb_depth 31 .. 48 (K = 7): the fused kernel on its 64-slot ring (round 2: two-kernel form); 49 and above: two kernels.
    Ragged batch, block lengths around the flush period, a NaN codeword ('soft'), against the oracle and the other paths."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import conv_encode_batch
    tr = make_trellis("k7_133_171")
    for B, nbits in ((70, 96), (5, 200), (129, 50), (64, 1024 if tb == 48 else 97)):
        rs = np.random.RandomState(100 * tb + nbits)
        coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
        if dtype == "hard":
            rx = np.where(rs.rand(*coded.shape) < 0.08, 1 - coded, coded)
        elif dtype == "soft":
            rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 2.0
            rx[B // 2, rs.randint(rx.shape[1])] = np.nan
        else:
            rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.8
        want = oracle.viterbi_decode(rx, tr, tb, dtype)
        got = _decode(rx, tr, tb, dtype, "cw!")
        note = _lib.last_kernel()
        assert ("64-slot ring" in note) == (tb <= 48) and ("viterbi_cw_acs_kernel" in note) == (tb > 48), note
        assert np.array_equal(got, want), (dtype, tb, B, nbits, "cw!")
        for path in ("cw2!", "wave"):
            assert np.array_equal(_decode(rx, tr, tb, dtype, path), want), (dtype, tb, B, nbits, path)


@pytest.mark.parametrize("gm,fmt", [([[0o133, 0o171]], "MSB"), ([[0o171, 0o133]], "MSB"), ([[0o133, 0o171]], "LSB"),
                                    ([[0o171, 0o133]], "LSB")])
def test_all_instantiated_generators(gpu, gm, fmt):
    from commpy_amd.channelcoding import Trellis, conv_encode_batch
    try:
        tr = Trellis(np.array([6]), np.array(gm), polynomial_format=fmt)
    except Exception:
        pytest.skip("polynomial format not supported by the host Trellis")
    rs = np.random.RandomState(7)
    coded = conv_encode_batch(rs.randint(0, 2, (150, 200)), tr).astype(float)
    rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 2.2
    got = _decode(rx, tr, None, "soft", "cw!")
    assert np.array_equal(got, oracle.viterbi_decode(rx, tr, None, "soft"))


@pytest.mark.parametrize("dtype", ["hard", "soft", "unquantized"])
@pytest.mark.parametrize("mem", [2, 3, 4, 5, 6])
def test_table_driven_codes(gpu, dtype, mem):
    """Any rate-1/2 shift-register code of 4 .. 64 states whose generators both tap the input and the oldest register bit takes the
    fused kernel with a run-time code table (csrc/viterbi_cw.hip, G0 = G1 = 0) at its default traceback depth: random generator
    pairs, ragged batches, a NaN codeword ('soft'), against the oracle and the state-per-lane kernels; a code without both end
    taps is refused."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import Trellis, conv_encode_batch
    rs = np.random.RandomState(31 + mem)
    ends = (1 << mem) | 1
    seen = 0
    for trial in range(5 if mem > 2 else 2):
        g0, g1 = (int(ends | (rs.randint(0, 1 << (mem - 1)) << 1)) for _ in range(2))   # both end bits set, middle taps random
        if g0 == g1:
            g1 ^= 2 if mem > 1 else 0
        if g0 == g1:
            continue
        tr = Trellis(np.array([mem]), np.array([[g0, g1]]))
        for B, nbits in ((70, 97), (5, 200), (129, 40)):
            coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
            if dtype == "hard":
                rx = np.where(rs.rand(*coded.shape) < 0.08, 1 - coded, coded)
            elif dtype == "soft":
                rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 2.0
                rx[B // 2, rs.randint(rx.shape[1])] = np.nan
            else:
                rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.8
            want = oracle.viterbi_decode(rx, tr, None, dtype)
            got = _decode(rx, tr, None, dtype, "cw!")
            note = _lib.last_kernel()
            assert "viterbi_cw_fused_kernel" in note, (note, oct(g0), oct(g1))
            seen += "table-driven" in note
            assert np.array_equal(got, want), (dtype, mem, oct(g0), oct(g1), B, nbits, note)
            assert np.array_equal(_decode(rx, tr, None, dtype, "wave"), want)
    assert seen > 0 or mem == 2                                       # (memory 2 has one such pair, (5,7): compiled in)
    # round 4: traceback depths below the default stay on the table-driven kernel (run-time hop count)
    if mem > 2:
        for tb in (2, 5 * mem // 2, 5 * mem - 1):
            B, nbits = 66, 120
            coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
            rx = {"hard": lambda: np.where(rs.rand(*coded.shape) < 0.08, 1 - coded, coded),
                  "soft": lambda: 4.0 * coded - 2 + rs.randn(*coded.shape) * 2.0,
                  "unquantized": lambda: 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.8}[dtype]()
            got = _decode(rx, tr, tb, dtype, "cw!")
            note = _lib.last_kernel()
            assert "viterbi_cw_fused_kernel" in note and ("runtime hops" in note), note
            assert np.array_equal(got, oracle.viterbi_decode(rx, tr, tb, dtype)), (dtype, mem, tb, note)
    tr = Trellis(np.array([mem]), np.array([[ends & ~1 | 2, ends]])) if mem > 1 else None   # first generator does not tap the oldest bit
    with pytest.raises(ValueError):
        _decode(np.zeros((2, 60)), tr, None, "hard", "cw!")


@pytest.mark.parametrize("dtype", ["hard", "soft", "unquantized"])
@pytest.mark.parametrize("name", ["t57", "k5_23_35"])
def test_small_trellises_on_the_fused_kernel(gpu, name, dtype):
    """K = 3 (5,7) -- BASELINE config 1 -- and K = 5 (23,35) at their default traceback depths: the fused kernel on its small ring;
    ragged batches, block lengths around the flush period, a NaN codeword ('soft'); shallower depths run the same kernel with a
    run-time hop count (round 4), deeper ones fall back to the other paths."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import conv_encode_batch
    tr = make_trellis(name)
    rs = np.random.RandomState(len(name) + len(dtype))
    for B, nbits in ((70, 64), (5, 200), (129, 97), (64, 1000)):
        coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
        if dtype == "hard":
            rx = np.where(rs.rand(*coded.shape) < 0.06, 1 - coded, coded)
        elif dtype == "soft":
            rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 1.5
            rx[B // 2, rs.randint(rx.shape[1])] = np.nan
        else:
            rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.7
        want = oracle.viterbi_decode(rx, tr, None, dtype)
        got = _decode(rx, tr, None, dtype, "cw!")
        assert "small ring" in _lib.last_kernel(), _lib.last_kernel()
        assert np.array_equal(got, want), (name, dtype, B, nbits)
        assert np.array_equal(_decode(rx, tr, None, dtype, "wave"), want)
    # round 4: depths BELOW the default stay on the small-ring kernel (run-time hop count); deeper windows have no instantiation
    default_tb = 5 * tr.total_memory
    for tb in (2, 3, default_tb // 2, default_tb - 1):
        B, nbits = 67, 150
        coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
        rx = {"hard": lambda: np.where(rs.rand(*coded.shape) < 0.06, 1 - coded, coded),
              "soft": lambda: 4.0 * coded - 2 + rs.randn(*coded.shape) * 1.5,
              "unquantized": lambda: 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.7}[dtype]()
        got = _decode(rx, tr, tb, dtype, "cw!")
        note = _lib.last_kernel()
        assert "small ring" in note and "runtime hops" in note, note
        assert np.array_equal(got, oracle.viterbi_decode(rx, tr, tb, dtype)), (name, dtype, tb)
    with pytest.raises(ValueError):
        _decode(np.zeros((2, 80)), tr, default_tb + 1, "hard", "cw!")


def test_generator_pairs_vs_live_reference(gpu):
    """tests/golden/viterbi_pairs.npz: ten generator pairs of memory 2 .. 6 (table-driven, small-ring and compiled-in fused kernels,
    the 64-slot ring at depth 40) decoded by the LIVE reference -- through the forced codeword path and the state-per-lane kernels."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import Trellis
    from test_oracle_golden import pair_cases
    g, cases = pair_cases()
    seen = set()
    for key, mem, g0, g1, dtype, tb in cases:
        tr = Trellis(np.array([mem]), np.array([[g0, g1]]))
        got = _decode(g[key + "__rx"], tr, tb, dtype, "cw!")
        note = _lib.last_kernel()
        seen.add("table" if "table-driven" in note else "small" if "small ring" in note else "deep" if "64-slot" in note else "compiled")
        assert np.array_equal(got, g[key + "__dec"]), (key, note)
        assert np.array_equal(_decode(g[key + "__rx"], tr, tb, dtype, "wave"), g[key + "__dec"]), key
    assert seen >= {"table", "deep", "compiled"}, seen              # ((5,7) and (23,35) on the small ring: the t57 / k5_23_35 goldens)


def test_table_driven_code_full_batch_default_dispatch(gpu):
    """A 64-state code without a compiled-in instantiation, (135,147), as a batch that fills the chip: the default dispatch takes the
    table-driven fused kernel; all 40 000 codewords equal the state-per-lane kernels, the first / last 700 the oracle."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import Trellis, conv_encode_batch, viterbi_decode
    tr = Trellis(np.array([6]), np.array([[0o135, 0o147]]))
    rs = np.random.RandomState(77)
    B = 40000
    coded = conv_encode_batch(rs.randint(0, 2, (B, 256)).astype(np.uint8), tr).astype(np.float64)
    rx = 4.0 * coded - 2 + rs.standard_normal(coded.shape) * 1.6
    got = viterbi_decode(rx, tr, None, "soft")
    assert "table-driven" in _lib.last_kernel(), _lib.last_kernel()
    assert np.array_equal(got, _decode(rx, tr, None, "soft", "wave"))
    for lo in (0, B - 700):
        assert np.array_equal(got[lo:lo + 700], oracle.viterbi_decode_mt(rx[lo:lo + 700], tr, None, "soft"))


def test_continuous_termination_and_short_windows(gpu):
    """'cont' streams (no tail), tb_depth larger than the block, L smaller than the window."""
    from commpy_amd.channelcoding import conv_encode
    tr = make_trellis("k7_133_171")
    rs = np.random.RandomState(3)
    for nbits, tb in ((40, None), (12, 30), (100, 10), (64, 35)):
        msgs = rs.randint(0, 2, (9, nbits))
        coded = np.stack([conv_encode(m, tr, "cont") for m in msgs]).astype(float)
        rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 1.5
        got = _decode(rx, tr, tb, "soft", "cw!")
        assert np.array_equal(got, _decode(rx, tr, tb, "soft", "wave")), (nbits, tb)
        steps = nbits + 6 - 1
        if (tb or min(30, nbits)) - 1 <= steps:      # otherwise the reference never traces back (deviations)
            assert np.array_equal(got, oracle.viterbi_decode(rx, tr, tb, "soft")), (nbits, tb)


def test_full_size_default_dispatch_equals_wave_kernels(gpu):
    """BASELINE config-2 size plus a partial round (the default dispatch sends one full round of 65536 codewords through
    the fused kernel and the last 5000 through the wave kernels): same bits as every forced path for all codewords; slices
    against the oracle; the noiseless batch decodes to the messages."""
    from commpy_amd.channelcoding import conv_encode_batch
    tr = make_trellis("k7_133_171")
    rs = np.random.RandomState(10)
    B = 65536 + 5000
    msgs = rs.randint(0, 2, (B, 1024))
    coded = conv_encode_batch(msgs, tr).astype(np.float64)
    clean = _decode(4.0 * coded - 2, tr, None, "soft", "cw!")
    assert np.array_equal(clean[:, :1024], msgs)
    rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 1.6
    auto = _decode(rx, tr, None, "soft", "auto")
    assert np.array_equal(auto, _decode(rx, tr, None, "soft", "cw!"))
    assert np.array_equal(auto, _decode(rx, tr, None, "soft", "cw2!"))
    assert np.array_equal(auto, _decode(rx, tr, None, "soft", "wave"))
    sel = np.r_[0:48, 65536 - 24:65536 + 24, B - 48:B]
    assert np.array_equal(auto[sel], oracle.viterbi_decode(rx[sel], tr, None, "soft"))
    assert 0 < np.mean(auto[:, :1024] != msgs) < 0.05


def test_remainder_runs_beside_the_round(gpu):
    """Round 6: a 'soft' batch of one full round plus a remainder (65 536 + 3 000 codewords, K = 7, default depth) -- the round takes the
    32-slot ring stored once and the state-per-lane remainder is issued on the library's side stream beside it (cpx_last_kernel says
    so); the bits equal the state-per-lane kernels' on every codeword and the oracle's on the first / last ones, the noiseless batch
    decodes to the messages, and two host threads doing this at the same time (thread-private fork / join events, one shared side
    stream) get the same bits as one thread alone."""
    import threading
    from commpy_amd import _lib
    from commpy_amd.channelcoding import conv_encode_batch, viterbi_decode
    import os
    if os.environ.get("CPX_VITERBI_OVERLAP", "1")[:1] == "0":
        pytest.skip("overlap switched off in the environment")
    tr = make_trellis("k7_133_171")
    B, nbits = 65536 + 3000, 96
    out = {}

    def work(seed):
        rs = np.random.RandomState(seed)
        msgs = rs.randint(0, 2, (B, nbits))
        coded = conv_encode_batch(msgs, tr).astype(np.float64)
        rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 1.5
        got = viterbi_decode(rx, tr, None, "soft")
        out[seed] = (rx, got, _lib.last_kernel(), msgs)
    work(1)
    rx, got, note, msgs = out[1]
    assert "ring stored once" in note and "beside the round" in note, note
    assert np.array_equal(got, _decode(rx, tr, None, "soft", "wave"))
    for lo in (0, 65536 - 200, B - 400):
        assert np.array_equal(got[lo:lo + 400], oracle.viterbi_decode_mt(rx[lo:lo + 400], tr, None, "soft"))
    clean = viterbi_decode(4.0 * conv_encode_batch(msgs, tr).astype(np.float64) - 2, tr, None, "soft")
    assert np.array_equal(clean[:, :nbits], msgs)
    single = {seed: None for seed in (2, 3)}
    for seed in single:
        work(seed)
        single[seed] = out[seed][1].copy()
    th = [threading.Thread(target=work, args=(seed,)) for seed in single]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for seed in single:
        assert np.array_equal(out[seed][1], single[seed]), seed
