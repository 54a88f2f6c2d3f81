"""Round 4: the reference's argument domain BEYOND the specialised kernels, through the general paths of the library --
viterbi_generic.hip (more than 128 states, k > 2, n > 6, traceback windows above the LDS ring), bcjr_exact.hip as the only
MAP / turbo path above 16 states, ldpc_exact_kernel for checks of more than 32 edges, the HBM-table demodulator above 256
points -- against LIVE-reference fixtures (tests/golden/general.npz; generator: tests/golden/make_golden.py gen_general)
and against the CPU oracle on randomised cases.  Bit-exact for integer outputs, 1e-5 on LLRs."""
import numpy as np
import pytest

import oracle
from helpers import GeneralTrellis, Perm, golden, make_trellis, viterbi_valid_bits

pytestmark = pytest.mark.gpu


def _names(kind):
    return [str(k) for k in golden("general")[kind]]


def _close_with_pattern(a, ref, tol, what):
    fin = np.isfinite(ref)
    assert np.array_equal(fin, np.isfinite(a)), what
    assert np.array_equal(a[~fin], ref[~fin], equal_nan=True), what
    assert np.max(np.abs(a[fin] - ref[fin]), initial=0.0) <= tol, (what, np.max(np.abs(a[fin] - ref[fin])))


@pytest.mark.parametrize("key", _names("vit_names"))
def test_viterbi_general_vs_live_reference(gpu, key):
    from commpy_amd import _lib
    from commpy_amd.channelcoding import viterbi_decode
    g = golden("general")
    tag, dtype, tb, _ = key.split("|")
    tr = GeneralTrellis(tag)                                       # a duck-typed trellis, like the one the reference was given
    rx = g[key + "__rx"]
    tbd = None if tb == "None" else int(tb)
    dec = viterbi_decode(rx, tr, tbd, dtype)
    # (K = 7 with a window of 600 steps still fits the LDS ring of the state-per-lane kernel: that case checks the general
    #  kernel by forcing it below)
    # K = 9 (256 states, k = 1) runs four states per lane on the wide kernel; everything else here is the general kernel's
    want_path = {"k7_tb": None, "k9_561_753": "wide"}.get(tag, "general")
    assert want_path is None or want_path in _lib.viterbi_last_path(), _lib.last_kernel()
    nv = viterbi_valid_bits(rx.shape[1], tr)
    assert dec.dtype == np.int64 and np.array_equal(dec[:, :nv], g[key + "__dec"][:, :nv])
    one = viterbi_decode(rx[0], tr, tbd, dtype)                    # 1-D call, the reference's own shape
    assert np.array_equal(one[:nv], g[key + "__dec"][0, :nv])
    try:
        _lib.viterbi_set_path("general")
        forced = viterbi_decode(rx, tr, tbd, dtype)
        assert _lib.viterbi_last_path() == "general"
    finally:
        _lib.viterbi_set_path(None)
    assert np.array_equal(forced[:, :nv], g[key + "__dec"][:, :nv])


@pytest.mark.parametrize("name", ["t57", "k2_default", "k2_rsc_matrix", "k7_133_171", "rsc_legacy_8", "r13_k4", "k8_247_371"])
@pytest.mark.parametrize("dtype", ["hard", "soft", "unquantized"])
def test_general_kernel_equals_specialised_kernels(gpu, name, dtype):
    """The general kernel forced onto trellises the specialised kernels serve: same bits, all batch sizes / depths incl. NaN."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import conv_encode_batch, viterbi_decode
    tr = make_trellis(name)
    import zlib
    rs = np.random.RandomState(zlib.crc32((name + dtype).encode()))
    for B, nbits, tb in ((1, 24, None), (37, 150, None), (5, 300, 7), (3, 40, 90), (130, 64, 12)):
        nbits -= nbits % tr.k
        coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
        if dtype == "hard":
            rx = np.where(rs.rand(*coded.shape) < 0.08, 1 - coded, coded)
        elif dtype == "soft":
            rx = (4.0 * coded - 2) + rs.randn(*coded.shape) * 2.5
            rx[rs.rand(*rx.shape) < 0.01] = np.inf
            if B > 4:
                rx[2, rx.shape[1] // 2] = np.nan                   # poisons the codeword from that step on (convcode.py:719)
        else:
            rx = (2.0 * coded - 1) + rs.randn(*coded.shape)
        ref = viterbi_decode(rx, tr, tb, dtype)
        try:
            _lib.viterbi_set_path("general")
            dec = viterbi_decode(rx, tr, tb, dtype)
            assert _lib.viterbi_last_path() == "general"
        finally:
            _lib.viterbi_set_path(None)
        assert np.array_equal(dec, ref), (name, dtype, B, nbits, tb)
        want = oracle.viterbi_decode(rx, tr, tb, dtype)
        if tb is None or tb - 1 <= int((want.shape[1] + tr.total_memory) / tr.k) - 1:   # else the reference returns np.empty memory
            assert np.array_equal(dec, want), (name, dtype, B, nbits, tb)


def test_k9_batch_on_the_wide_kernel_vs_oracle_and_general_kernel(gpu):
    """K = 9 (561,753), the standard 256-state code: 300 codewords of 200 bits, soft with +-inf and one NaN codeword, through
    the four-states-per-lane kernel (and its NaN redo) = the general kernel = the oracle."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import Trellis, conv_encode_batch, viterbi_decode
    tr = Trellis(np.array([8]), np.array([[0o561, 0o753]]))
    rs = np.random.RandomState(9)
    B, nbits = 300, 200
    coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
    for dtype in ("soft", "hard", "unquantized"):
        if dtype == "soft":
            rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 2.2
            rx[rs.rand(*rx.shape) < 0.003] = np.inf
            rx[7, 130] = np.nan
        elif dtype == "hard":
            rx = np.where(rs.rand(*coded.shape) < 0.07, 1 - coded, coded)
        else:
            rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.9
        for tb in (None, 17):
            dec = viterbi_decode(rx, tr, tb, dtype)
            assert _lib.viterbi_last_path() == "wide", _lib.last_kernel()
            want = oracle.viterbi_decode(rx[:40], tr, tb, dtype)
            assert np.array_equal(dec[:40], want), (dtype, tb)
            try:
                _lib.viterbi_set_path("general")
                gen = viterbi_decode(rx, tr, tb, dtype)
            finally:
                _lib.viterbi_set_path(None)
            assert np.array_equal(dec, gen), (dtype, tb)


def test_viterbi_general_random_large_codes_vs_oracle(gpu):
    """Random feed-forward codes of memory 7 .. 10 and random k = 3 / n up to 12 codes against the oracle."""
    from commpy_amd.channelcoding import Trellis, conv_encode_batch, viterbi_decode
    rs = np.random.RandomState(77)
    for it in range(10):
        if it % 2 == 0:
            m = int(rs.randint(7, 11))
            n = int(rs.randint(2, 4))
            gm = rs.randint(1, 2 ** (m + 1), (1, n))
            gm[0, 0] |= 1 | (1 << m)
            tr = Trellis(np.array([m]), gm)
        else:
            mem = rs.randint(1, 3, 3)
            n = int(rs.randint(4, 13))
            gm = rs.randint(0, 2 ** (int(mem.max()) + 1), (3, n))
            for i in range(3):
                gm[i, i] |= 1
            tr = Trellis(mem, gm)
        try:
            tr._device_handle()
        except ValueError:                                         # irregular in-degree: the reference breaks too
            continue
        B, nbits = int(rs.randint(1, 9)), int(rs.randint(30, 120))
        nbits -= nbits % tr.k
        coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
        dtype = ("hard", "soft", "unquantized")[it % 3]
        rx = {"hard": lambda: np.where(rs.rand(*coded.shape) < 0.1, 1 - coded, coded),
              "soft": lambda: (4.0 * coded - 2) + rs.randn(*coded.shape) * 3.0,
              "unquantized": lambda: (2.0 * coded - 1) + rs.randn(*coded.shape) * 1.2}[dtype]()
        L = int(rx.shape[1] * tr.k / tr.n)
        T = int((L + tr.total_memory) / tr.k) - 1
        tbd = None if min(5 * tr.total_memory, L) - 1 <= T else max(2, T // 2)   # the default would never trace back (np.empty memory)
        dec = viterbi_decode(rx, tr, tbd, dtype)
        nv = viterbi_valid_bits(rx.shape[1], tr)
        assert np.array_equal(dec[:, :nv], oracle.viterbi_decode(rx, tr, tbd, dtype)[:, :nv]), (it, dtype)


@pytest.mark.parametrize("mem", [11, 12, 13])
def test_viterbi_thousands_of_states_vs_oracle(gpu, mem):
    """2048 states (path metrics still in LDS), 4096 and 8192 (metrics in the HBM scratch; 128 ballot words per decision plane)."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import Trellis, conv_encode_batch, viterbi_decode
    rs = np.random.RandomState(mem)
    gm = rs.randint(1, 2 ** (mem + 1), (1, 2))
    gm[0] |= 1 | (1 << mem)
    tr = Trellis(np.array([mem]), gm)
    B, nbits = 3, 70
    coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr).astype(float)
    rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 2.5
    for tb in (None, 20):
        dec = viterbi_decode(rx, tr, tb, "soft")
        assert "general" in _lib.viterbi_last_path()
        assert np.array_equal(dec, oracle.viterbi_decode(rx, tr, tb, "soft")), (mem, tb)


@pytest.mark.parametrize("key", _names("map_names"))
def test_map_decode_32_and_64_states_vs_live_reference(gpu, key):
    from commpy_amd import _lib
    from commpy_amd.channelcoding import map_decode
    g = golden("general")
    tr = GeneralTrellis(key.split("|")[0])
    L, bits = map_decode(g[key + "__sys"], g[key + "__par"], tr, float(g[key + "__nv"]), g[key + "__lint"], "decode")
    assert "map_exact_kernel" in _lib.last_kernel()
    _close_with_pattern(L, g[key + "__L"], 1e-5, key)
    sure = np.abs(g[key + "__L"]) > 1e-5
    assert np.array_equal(bits[sure], g[key + "__bits"][sure])


@pytest.mark.parametrize("key", _names("turbo_names"))
def test_turbo_decode_32_and_64_states_vs_live_reference(gpu, key):
    from commpy_amd.channelcoding import turbo_decode
    g = golden("general")
    tr = GeneralTrellis(key.split("|")[0])
    dec = turbo_decode(g[key + "__sys"], g[key + "__p1"], g[key + "__p2"], tr, float(g[key + "__nv"]), int(g[key + "__iters"]),
                       Perm(g[key + "__perm"]))
    assert np.array_equal(dec, g[key + "__dec"])


def test_map_decode_batch_many_states_vs_oracle(gpu):
    """A batch (more codewords than lanes of a wavefront) of a 128-state RSC code: every lane / scratch column is exercised."""
    from commpy_amd.channelcoding import Trellis, conv_encode_batch, map_decode
    tr = Trellis(np.array([7]), np.array([[1, 0o345]]), np.array([[0o237]]), 'rsc')
    rs = np.random.RandomState(5)
    B, N = 70, 40
    coded = conv_encode_batch(rs.randint(0, 2, (B, N)), tr, 'cont')
    sy = 2.0 * coded[:, 0::2] - 1 + 0.8 * rs.randn(B, N)
    pa = 2.0 * coded[:, 1::2] - 1 + 0.8 * rs.randn(B, N)
    li = rs.randn(B, N)
    L, bits = map_decode(sy, pa, tr, 0.64, li, 'decode')
    for b in (0, 1, 63, 64, 69):
        Lo, bo = oracle.map_decode(sy[b], pa[b], tr, 0.64, li[b], 'decode')
        assert np.max(np.abs(L[b] - Lo)) < 1e-9
        assert np.array_equal(bits[b], bo)


@pytest.mark.parametrize("key", _names("ldpc_names"))
def test_ldpc_check_degree_40_vs_live_reference(gpu, key):
    import scipy.sparse as sp
    from commpy_amd import _lib
    from commpy_amd.channelcoding import ldpc_bp_decode
    g = golden("general")
    H = g["ldpc_H"]
    params = {"n_vnodes": H.shape[1], "n_cnodes": H.shape[0], "parity_check_matrix": sp.csc_matrix(H)}
    alg, its = key.split("|")[1], int(key.split("|")[3])
    llr = g[key + "__llr"].copy()
    dec, out = ldpc_bp_decode(llr, params, alg, its)
    assert "ldpc_exact_kernel" in _lib.last_kernel()
    assert dec.shape == g[key + "__dec"].shape and dec.dtype == np.int8
    if alg == "MSA":
        assert np.array_equal(out, g[key + "__out"])
    else:
        assert np.max(np.abs(out - g[key + "__out"])) <= 1e-5
    assert np.array_equal(dec, g[key + "__dec"])
    big = np.full(H.shape[1], 700.0)                               # the in-place clip of ldpc.py:186 on this path too
    ldpc_bp_decode(big, params, alg, 1)
    assert np.all(big == 500.0)


@pytest.mark.parametrize("tag", _names("demod_names"))
def test_demod_large_constellations_vs_live_reference(gpu, tag):
    from commpy_amd.modulation import Modem
    g = golden("general")
    md = Modem(g[tag + "__cst"], reorder_as_gray=False)            # the stored table is the reference modem's, already re-indexed
    _close_with_pattern(md.demodulate(g[tag + "__y"], "soft", float(g[tag + "__nv"])), g[tag + "__soft"], 1e-5, tag)
    hard = md.demodulate(g[tag + "__y"], "hard")
    assert hard.dtype == np.int8 and np.array_equal(hard, g[tag + "__hard"])
    # modulate (device table gather above 256 points) round trip
    from commpy_amd.devicelink import modulate_gpu
    bits = np.random.RandomState(3).randint(0, 2, 40 * md.num_bits_symbol)
    sym = modulate_gpu(md, bits)
    assert np.array_equal(sym, md.modulate(bits))
    assert np.array_equal(md.demodulate(sym, "hard"), bits)


# ---- soft demodulator: four exponentials per axis (geometric progression) vs one per level vs the oracle ------------------------
@pytest.mark.parametrize("m", [4, 64, 256])
def test_soft_demod_progression_form_equals_plain_form_and_oracle(gpu, m):
    """Square QAM of 64 points and more takes the progression form (demod.hip, GP), QPSK / 4-QAM the closed form (no exp at all).  From -5 dB to 40 dB Es/N0, with outliers
    far outside the constellation, scaled and shifted constellations: same non-finite pattern as the oracle, values within 1e-9
    of it (the contract is 1e-5) and within 1e-9 of the plain form."""
    from commpy_amd import _lib
    from commpy_amd.modulation import Modem, QAMModem
    base = QAMModem(m)
    rs = np.random.RandomState(m)
    worst = 0.0
    for scale, shift in ((1.0, 0.0), (1.0 / np.sqrt(base.Es), 0.0), (0.37, 0.11 - 0.07j)):
        md = Modem(base.constellation * scale + shift, reorder_as_gray=False)
        es = np.mean(np.abs(md.constellation - shift) ** 2)
        for snr_db in (-5.0, 8.0, 18.0, 27.0, 33.0, 40.0):
            N0 = es / 10 ** (snr_db / 10)
            ns = 3000
            y = md.constellation[rs.randint(0, md.m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
            y[::97] *= 6.0                                         # far outside the constellation
            y[5] = complex(np.inf, 0.0)
            y[6] = complex(np.nan, 1.0)
            soft = md.demodulate(y, "soft", N0)
            assert ",gp" in _lib.last_kernel(), _lib.last_kernel()
            try:
                _lib.demod_set_path("plain")
                plain = md.demodulate(y, "soft", N0)
                assert ",gp" not in _lib.last_kernel()
            finally:
                _lib.demod_set_path(None)
            want = oracle.demodulate(md.constellation, y, "soft", N0)
            for got in (soft, plain):
                assert np.array_equal(np.isfinite(got), np.isfinite(want)), (scale, snr_db)
                assert np.array_equal(np.isnan(got), np.isnan(want)), (scale, snr_db)
                inf = np.isinf(want)
                assert np.array_equal(got[inf], want[inf])
            fin = np.isfinite(want)
            dev = float(np.max(np.abs(soft[fin] - want[fin]), initial=0.0))       # (QPSK at 40 dB: no finite LLR is left)
            worst = max(worst, dev)
            assert dev < 1e-9, (scale, snr_db, dev)
            assert np.max(np.abs(soft[fin] - plain[fin]), initial=0.0) < 1e-9
    print("soft demod progression form, QAM-%d: max |LLR - oracle| = %.3g" % (m, worst))


def test_soft_demod_progression_needs_gray_equally_spaced_axes(gpu):
    """A separable constellation whose levels are NOT equally spaced keeps the plain form."""
    from commpy_amd import _lib
    from commpy_amd.modulation import Modem, QAMModem
    c = QAMModem(64).constellation.copy()
    c = np.sign(c.real) * np.abs(c.real) ** 1.2 + 1j * c.imag     # warped real axis: still label = (a << 3) | b
    md = Modem(c, reorder_as_gray=False)
    rs = np.random.RandomState(1)
    y = c[rs.randint(0, 64, 500)] + 0.3 * (rs.randn(500) + 1j * rs.randn(500))
    soft = md.demodulate(y, "soft", 0.2)
    assert "demod_soft_sep_kernel" in _lib.last_kernel() and ",gp" not in _lib.last_kernel()
    assert np.max(np.abs(soft - oracle.demodulate(c, y, "soft", 0.2))) < 1e-9
