"""GPU parity of the BCJR/turbo, LDPC-BP and demodulation kernels (through the C-ABI) against the
golden vectors from the live reference and the CPU oracle.

Tolerances (BASELINE.json north_star): decoded bits / dec_word / hard decisions bit-exact;
float LLR outputs within 1e-5 absolute."""
import ctypes

import numpy as np
import pytest

import oracle
from helpers import Perm, golden, ldpc_params, make_trellis

pytestmark = pytest.mark.gpu
TOL = 1e-5


# ------------------------------------------------------------------ BCJR / MAP
def test_map_decode_golden(gpu):
    from commpy_amd.channelcoding import map_decode
    g = golden("map_turbo")
    worst = 0.0
    for nm in g["map_names"]:
        key, tname, N, nv, lk, mode = str(nm).split("|")
        tr = make_trellis(tname)
        L, bits = map_decode(g[key + "__sys"], g[key + "__par"], tr, float(g[key + "__nv"]), g[key + "__lint"], mode)
        worst = max(worst, float(np.max(np.abs(L - g[key + "__L"]))))
        assert bits.dtype == np.int64
        # hard decisions may only differ where |L| is within the tolerance of 0
        diff = bits != g[key + "__bits"]
        assert not np.any(diff & (np.abs(g[key + "__L"]) > TOL)), nm
    assert worst < TOL, worst


def test_map_decode_batch_vs_oracle(gpu):
    from commpy_amd.channelcoding import map_decode
    rs = np.random.RandomState(5)
    for tname in ("rsc_legacy_4", "rsc_legacy_8", "k5_23_35"):
        tr = make_trellis(tname)
        B, N = 37, 300
        s = rs.randn(B, N) * 1.2
        p = rs.randn(B, N) * 1.2
        li = rs.randn(B, N)
        L, bits = map_decode(s, p, tr, 0.9, li, "decode")
        assert L.shape == (B, N) and bits.shape == (B, N)
        for b in (0, 17, 36):
            Lo, bo = oracle.map_decode(s[b], p[b], tr, 0.9, li[b], "decode")
            assert np.max(np.abs(L[b] - Lo)) < TOL
            assert not np.any((bits[b] != bo) & (np.abs(Lo) > TOL))


def test_turbo_decode_golden(gpu):
    from commpy_amd.channelcoding import turbo_decode
    g = golden("map_turbo")
    for nm in g["turbo_names"]:
        key, tname, N, nv, iters = str(nm).split("|")
        tr = make_trellis(tname)
        dec = turbo_decode(g[key + "__sys"], g[key + "__p1"], g[key + "__p2"], tr, float(g[key + "__nv"]), int(iters),
                           Perm(g[key + "__perm"]))
        assert dec.dtype == np.int64
        assert np.array_equal(dec, g[key + "__dec"]), nm


def test_turbo_config3_batch_roundtrip(gpu):
    """Config-3 shape (rate 1/3, N=1024, 6 iterations, random interleaver): batch of encoded blocks over
    AWGN; outputs equal the oracle's on a sample and the messages at this SNR."""
    from commpy_amd.channelcoding import RandInterlv, turbo_decode, turbo_encode
    tr = make_trellis("rsc_legacy_4")
    N, B = 1024, 96
    il = RandInterlv(N, 1234)
    rs = np.random.RandomState(20)
    msgs = rs.randint(0, 2, (8, N))
    enc = [turbo_encode(m, tr, tr, il) for m in msgs]
    s = np.tile(np.stack([e[0] for e in enc]), (B // 8, 1)) * 2.0 - 1
    p1 = np.tile(np.stack([e[1] for e in enc]), (B // 8, 1)) * 2.0 - 1
    p2 = np.tile(np.stack([e[2][:N] for e in enc]), (B // 8, 1)) * 2.0 - 1
    nv = 1 / (2 * (1.0 / 3) * 10 ** (1.5 / 10.0))
    nrs = np.random.RandomState(21)
    s, p1, p2 = (a + np.sqrt(nv) * nrs.randn(B, N) for a in (s, p1, p2))
    dec = turbo_decode(s, p1, p2, tr, nv, 6, il)
    assert dec.shape == (B, N)
    for b in (0, 41, 95):
        assert np.array_equal(dec[b], oracle.turbo_decode(s[b], p1[b], p2[b], tr, nv, 6, il))
    ber = np.mean(dec != np.tile(msgs, (B // 8, 1)))
    assert ber < 5e-3, ber


# ------------------------------------------------------------------ LDPC
CHAOTIC = {"l026"}   # +-500-clipped saturation stress: SPA is 1-ulp chaotic there; MSA still exact


def test_ldpc_golden(gpu):
    from commpy_amd.channelcoding import ldpc_bp_decode
    g = golden("ldpc")
    for nm in g["names"]:
        key, cname, nblk, alg, iters = str(nm).split("|")
        p = ldpc_params(cname)
        llr = g[key + "__llr"].copy()
        dec, out = ldpc_bp_decode(llr, p, alg, int(iters))
        assert dec.dtype == np.int8 and dec.shape == g[key + "__dec"].shape
        assert np.array_equal(llr, np.clip(g[key + "__llr"], -500, 500))      # in-place clip
        if key in CHAOTIC and alg == "SPA":
            assert np.all(np.isfinite(out))
            continue
        assert np.array_equal(dec, g[key + "__dec"]), nm
        assert np.max(np.abs(out - g[key + "__out"])) < TOL, (nm, np.max(np.abs(out - g[key + "__out"])))


def test_ldpc_noiseless_encode_decode(gpu):
    """test_ldpc.py:77-106: WiMax systematic encode -> noiseless +-1 'LLRs' -> both algorithms recover the message."""
    from commpy_amd.channelcoding import ldpc_bp_decode, triang_ldpc_systematic_encode
    g = golden("ldpc")
    p = ldpc_params("wimax1440")
    msg = g["enc1440__msg"]
    coded = triang_ldpc_systematic_encode(msg, p)
    assert np.array_equal(coded, g["enc1440__coded"])
    sym = np.where(coded == 1, -1.0, 1.0).reshape(-1, order="F")
    for alg in ("SPA", "MSA"):
        dec, out = ldpc_bp_decode(sym.copy(), p, alg, 10)
        assert np.array_equal(dec, g["enc1440__dec_" + alg])
        assert np.max(np.abs(out - g["enc1440__out_" + alg])) < TOL
        assert np.array_equal(dec[:720].reshape(-1, order="F")[:len(msg)], msg)


def test_ldpc_batch_vs_oracle(gpu):
    """802.11n-style (1944,1296) code, a ragged batch (B not a multiple of the block width), early exit
    bookkeeping: executed iterations, dec_word and out_llrs against the oracle."""
    from commpy_amd.channelcoding import ldpc_bp_decode
    p = ldpc_params("n1944")
    n, B = 1944, 70
    rs = np.random.RandomState(31)
    sigma = 1 / np.sqrt(10 ** (2.6 / 10.0) * (2.0 / 3) * 2)
    llr = 2.0 * (1.0 + sigma * rs.randn(B * n)) / sigma ** 2
    for alg, iters in (("MSA", 12), ("SPA", 12)):
        dec, out, its = ldpc_bp_decode(llr.copy(), p, alg, iters, return_iterations=True)
        do, oo, io = oracle.ldpc_bp_decode(llr.copy(), p, alg, iters, True)
        assert np.array_equal(its, io), alg
        assert np.array_equal(dec, do), alg
        # SPA: 2*atanh(x) near |x| -> 1 amplifies a 1-ulp libm difference by 1/(1-|x|): strict 1e-5 holds for
        # messages below ~26, above that the deviation stays relative (DESIGN.md "LDPC parity").
        assert np.all(np.abs(out - oo) <= TOL + 1e-6 * np.abs(oo)), (alg, np.max(np.abs(out - oo)))


def test_ldpc_compaction_mixed_convergence(gpu):
    """Blocks that converge at very different iterations (noiseless ... undecodable) interleaved in one batch:
    exercises freezing, on-device compaction of the working set (several moves) and retirement at the
    original block index.  Min-sum is pure add/compare arithmetic: out_llrs must be bit-identical."""
    from commpy_amd.channelcoding import ldpc_bp_decode
    p = ldpc_params("n1944")
    n, B = 1944, 333
    rs = np.random.RandomState(77)
    ebn0 = rs.choice([0.5, 2.0, 2.6, 3.2, 4.5, 30.0], size=B)
    sig = 1 / np.sqrt(10 ** (ebn0 / 10.0) * (2.0 / 3) * 2)
    llr = (2.0 * (1.0 + sig[:, None] * rs.randn(B, n)) / sig[:, None] ** 2).reshape(-1)
    for alg, iters in (("MSA", 25), ("SPA", 9)):
        dec, out, its = ldpc_bp_decode(llr.copy(), p, alg, iters, return_iterations=True)
        do, oo, io = oracle.ldpc_bp_decode(llr.copy(), p, alg, iters, True)
        assert len(set(io.tolist())) >= 5                      # the batch really is staggered
        assert np.array_equal(its, io), alg
        assert np.array_equal(dec, do), alg
        if alg == "MSA":
            assert np.array_equal(out, oo)
        else:
            # 2*atanh amplifies a last-ulp difference of its argument by 1/(1 - |x|): 1e-5 absolute is the bar up to
            # |LLR| = 26 (DESIGN.md "LDPC-SPA note"), relative 1e-5 above it -- no implementation, the reference's
            # own NumPy builds included, agrees more closely there
            dev, mag = np.abs(out - oo), np.abs(oo)
            assert np.all(dev[mag <= 26.0] <= TOL), np.max(dev[mag <= 26.0])
            assert np.all(dev[mag > 26.0] <= 1e-5 * mag[mag > 26.0]), np.max(dev[mag > 26.0])


def test_ldpc_bad_algorithm(gpu):
    from commpy_amd.channelcoding import ldpc_bp_decode
    with pytest.raises(NameError):
        ldpc_bp_decode(np.zeros(96), ldpc_params("gallager96"), "BP", 3)


# ------------------------------------------------------------------ demodulation
def _modems():
    from commpy_amd.modulation import Modem, PSKModem, QAMModem
    return {
        "qam4": QAMModem(4), "qam16": QAMModem(16), "qam64": QAMModem(64), "qam256": QAMModem(256),
        "psk2": PSKModem(2), "psk4": PSKModem(4), "psk8": PSKModem(8), "psk16": PSKModem(16),
        "custom4": Modem([1 + 1j, -1.2 + 0.8j, 0.3 - 1j, -1 - 1.5j]),
        "custom8_nogray": Modem(np.exp(1j * np.arange(8) * 2 * np.pi / 8) * np.array([1, 2, 1, 2, 1, 2, 1, 2]),
                                reorder_as_gray=False),
    }


def test_demod_golden(gpu):
    g = golden("demod")
    mods = _modems()
    for nm in g["names"]:
        key, mname, N0 = str(nm).split("|")
        md = mods[mname]
        hard = md.demodulate(g[key + "__y"], "hard")
        assert hard.dtype == np.int8 and np.array_equal(hard, g[key + "__hard"]), nm
        soft = md.demodulate(g[key + "__y"], "soft", float(g[key + "__N0"]))
        ref = g[key + "__soft"]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(soft), fin)
        assert np.max(np.abs(soft[fin] - ref[fin])) < TOL, nm


def test_demod_modulate_identity(gpu):
    """test_modulation.py:159-162: modulate -> hard demodulate is the identity for every bit pattern."""
    for name, md in _modems().items():
        nb = md.num_bits_symbol
        bits = ((np.arange(md.m)[:, None] >> np.arange(nb - 1, -1, -1)) & 1).reshape(-1)
        assert np.array_equal(md.demodulate(md.modulate(bits), "hard"), bits), name


def test_demod_large_vs_oracle(gpu):
    from commpy_amd.modulation import QAMModem
    md = QAMModem(64)
    rs = np.random.RandomState(3)
    ns = 100003
    bits = rs.randint(0, 2, ns * 6)
    N0 = md.Es / 10 ** (14 / 10.0)
    y = md.modulate(bits) + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
    soft = md.demodulate(y, "soft", N0)
    hard = md.demodulate(y, "hard")
    idx = np.r_[0:300, ns - 300:ns]
    so = oracle.demodulate(md.constellation, y[idx], "soft", N0)
    ho = oracle.demodulate(md.constellation, y[idx], "hard")
    sel = (idx[:, None] * 6 + np.arange(6)).reshape(-1)
    assert np.max(np.abs(soft[sel] - so)) < TOL
    assert np.array_equal(hard[sel], ho)
    # hard decision == sign of the soft LLR wherever the LLR is not tiny
    assert np.mean((soft > 0) == (hard == 1)) > 0.999


def test_demod_bad_type(gpu):
    from commpy_amd.modulation import QAMModem
    with pytest.raises(ValueError):
        QAMModem(4).demodulate(np.zeros(3, complex), "fuzzy")


@pytest.mark.parametrize("B", [1, 64, 65])
@pytest.mark.parametrize("iters", [0, 1, 2])
def test_ldpc_edge_sizes(gpu, B, iters):
    """Tile boundaries (one block, exactly one wavefront tile, one block more) and the iteration-count edge cases
    (0: hard decision of the clipped input; 1: no compaction scan at all) against the oracle, both algorithms."""
    from commpy_amd.channelcoding import ldpc_bp_decode
    p = ldpc_params("gallager96")
    n = int(p["n_vnodes"])
    rs = np.random.RandomState(B * 10 + iters)
    llr = rs.randn(B * n) * 2.0 + 1.0
    llr[::17] *= 400.0                                             # some values beyond the +-500 clip
    for alg in ("MSA", "SPA"):
        x = llr.copy()
        dec, out, its = ldpc_bp_decode(x, p, alg, iters, return_iterations=True)
        y = llr.copy()
        do, oo, io = oracle.ldpc_bp_decode(y, p, alg, iters, True)
        assert np.array_equal(x, y)                                # in-place clip of the caller's array (ldpc.py:186)
        assert np.array_equal(np.atleast_1d(its), np.atleast_1d(io))
        assert np.array_equal(dec, do)
        assert np.all(np.abs(out - oo) <= TOL + 1e-6 * np.abs(oo))


def test_ldpc_full_size_syndrome_property(gpu):
    """BASELINE config-4 per-GPU share (B = 32768 blocks of the (1944,1296) code, min-sum, <= 50 iterations):
    size-independent properties instead of an oracle run -- every block that stopped early satisfies all parity
    checks (H . dec = 0), a block that ran all iterations does not, out_llrs signs agree with dec_word, and a
    noiseless all-zero block is returned untouched after 0 iterations."""
    from commpy_amd.channelcoding import ldpc_bp_decode
    p = ldpc_params("n1944")
    n, B, iters = 1944, 32768, 50
    rs = np.random.RandomState(2024)
    sigma = 1 / np.sqrt(10 ** (3.0 / 10.0) * (2.0 / 3) * 2)
    llr = (2.0 * (1.0 + sigma * rs.standard_normal((B, n))) / sigma ** 2)
    llr[7] = 25.0                                                  # noiseless all-zero codeword
    dec, out, its = ldpc_bp_decode(llr.reshape(-1), p, "MSA", iters, return_iterations=True)
    assert dec.shape == (n, B) and out.shape == (n, B) and its.shape == (B,)
    assert its[7] == 0 and not dec[:, 7].any() and np.array_equal(out[:, 7], np.full(n, 25.0))
    ec, ev = oracle.ldpc_edges(p)
    syn = np.zeros((int(p["n_cnodes"]), B), np.int32)
    np.add.at(syn, ec, dec[ev].astype(np.int32))                   # H . dec over the integers, then mod 2
    unsat = (syn & 1).any(axis=0)
    assert not unsat[its < iters].any()                            # early exit <=> zero syndrome (ldpc.py:205)
    assert unsat[its == iters].sum() >= 0.5 * (its == iters).sum() # blocks that used every iteration mostly failed
    assert np.array_equal(dec == 1, np.signbit(out))               # dec_word = out_llrs < 0 (ldpc.py:248)
    assert 5.5 < its.mean() < 7.5 and (its < iters).mean() > 0.999


def test_demod_full_size_identity(gpu):
    """BASELINE config-4 demodulator size (32768 x 324 64-QAM symbols): modulate -> AWGN at high SNR -> hard
    demodulation returns the bits, and the sign of every soft LLR (log P1/P0) agrees with them."""
    from commpy_amd.modulation import QAMModem
    md = QAMModem(64)
    rs = np.random.RandomState(64)
    nsym = 32768 * 324
    bits = rs.randint(0, 2, nsym * 6).astype(np.int8)
    N0 = md.Es / 10 ** (32 / 10.0)
    y = md.modulate(bits)
    y = y + np.sqrt(N0 / 2) * (rs.standard_normal(nsym) + 1j * rs.standard_normal(nsym))
    hard = md.demodulate(y, "hard")
    assert hard.dtype == np.int8 and np.array_equal(hard, bits)
    soft = md.demodulate(y, "soft", N0)
    assert soft.shape == (nsym * 6,) and np.all(np.isfinite(soft) | np.isinf(soft))
    assert np.array_equal(soft > 0, bits == 1)


# ------------------------------------------------------------------ BCJR outside the reference's representable range
def _pattern_equal(a, b):
    """Same NaN positions, same +-inf positions (and signs)."""
    return (np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isposinf(a), np.isposinf(b)) and
            np.array_equal(np.isneginf(a), np.isneginf(b)))


@pytest.mark.parametrize("tname", ["rsc_legacy_4", "rsc_legacy_8"])
def test_map_decode_extreme_regimes_follow_the_reference(gpu, tname):
    """Promoted from scripts/micro/map_extreme.py.  Symbol amplitudes of 5 - 20 at sigma^2 <= 0.1 and priors of |L| up to 200:
    the terms of the reference's absolute-scale recursion underflow (turbo.py:62-76, :238-240), a column sum becomes 0 and the
    LLRs NaN, or app0 = 0 and the LLR +-inf.  The fast kernels flag every codeword for which that can happen and a literal
    absolute-scale kernel decodes it again (csrc/bcjr_exact.hip): NaN / +-inf pattern identical to the oracle (= the
    reference, checked on these regimes), finite values within 1e-5 + 1e-9 |L|, decisions equal."""
    from commpy_amd.channelcoding import map_decode
    tr = make_trellis(tname)
    rs = np.random.RandomState(7)
    n_nonfinite = 0
    for amp in (1.0, 5.0, 20.0):
        for nv in (0.02, 0.1, 1.0):
            for lsc in (0.0, 5.0, 60.0):
                B, N = 6, int(rs.randint(5, 120))
                s_ = (rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.5) * amp
                p_ = (rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.5) * amp
                L = rs.randn(B, N) * lsc
                Le, bits = map_decode(s_, p_, tr, nv, L, "decode")
                for b in range(B):
                    Lo, bo = oracle.map_decode(s_[b], p_[b], tr, nv, L[b], "decode")
                    key = (tname, amp, nv, lsc, b)
                    assert _pattern_equal(Le[b], Lo), key
                    fin = np.isfinite(Lo)
                    n_nonfinite += int(np.sum(~fin))
                    assert np.all(np.abs(Le[b][fin] - Lo[fin]) <= TOL + 1e-9 * np.abs(Lo[fin])), key
                    assert not np.any((bits[b] != bo) & ~(np.abs(Lo) <= TOL)), key      # NaN / inf positions included
    assert n_nonfinite > 500                                       # the regimes really are outside the reference's range


def test_map_decode_nonfinite_inputs_follow_the_reference(gpu):
    """NaN / +-inf among the received values or the a-priori LLRs: e^inf = inf gives p0 = 0, p1 = 1 (turbo.py:239-240); a NaN
    poisons the column sums from its step on, in both directions."""
    from commpy_amd.channelcoding import map_decode
    tr = make_trellis("rsc_legacy_4")
    rs = np.random.RandomState(8)
    B, N = 8, 60
    s_ = rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.7
    p_ = rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.7
    L = rs.randn(B, N) * 2
    s_[1, 17] = np.nan
    p_[2, 3] = np.inf
    L[3, 30] = np.inf
    L[4, 31] = -np.inf
    L[5, 0] = np.nan
    L[6, 10] = 800.0
    L[6, 11] = -800.0
    Le, bits = map_decode(s_, p_, tr, 0.5, L, "decode")
    for b in range(B):
        Lo, bo = oracle.map_decode(s_[b], p_[b], tr, 0.5, L[b], "decode")
        assert _pattern_equal(Le[b], Lo), b
        fin = np.isfinite(Lo)
        assert np.all(np.abs(Le[b][fin] - Lo[fin]) <= TOL + 1e-9 * np.abs(Lo[fin])), b
        assert not np.any((bits[b] != bo) & ~(np.abs(Lo) <= TOL)), b


def test_turbo_decode_extreme_regimes_follow_the_reference(gpu):
    """turbo_decode where the reference's MAP passes underflow (high amplitude at low noise variance, huge L_int): decoded bits
    equal the oracle's -- the flagged codewords go through the absolute-scale turbo kernel (csrc/bcjr_exact.hip)."""
    from commpy_amd.channelcoding import RandInterlv, turbo_decode
    tr = make_trellis("rsc_legacy_4")
    rs = np.random.RandomState(9)
    n = 0
    for amp, nv, lsc in ((5.0, 0.02, 0.0), (20.0, 0.1, 0.0), (1.0, 0.1, 60.0), (5.0, 1.0, 5.0), (1.0, 0.004, 0.0)):
        B, N = 20, int(rs.randint(40, 200))
        il = RandInterlv(N, 77)
        s_, p1, p2 = ((rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.5) * amp for _ in range(3))
        L = rs.randn(B, N) * lsc if lsc else None
        for iters in (1, 3):
            dec = turbo_decode(s_, p1, p2, tr, nv, iters, il, L)
            for b in range(0, B, 3):
                want = oracle.turbo_decode(s_[b], p1[b], p2[b], tr, nv, iters, il, None if L is None else L[b])
                assert np.array_equal(dec[b], want), (amp, nv, lsc, iters, b)
                n += 1
    assert n >= 60


@pytest.mark.parametrize("m,snr_db", [(16, 10.0), (64, 8.0), (64, 24.0), (256, 14.0)])
def test_soft_demod_ragged_sizes_path_modes_and_bounds(gpu, m, snr_db):
    """The separable kernel stores a wave's 64 consecutive symbols as one contiguous run through an LDS transpose, works on a thread's
    symbols as a software pipeline and evaluates exp / log from tables (round 5): every size around the wave and block boundaries, in
    all three path modes, against the oracle on every symbol -- and nothing may be written behind the Ns * nb values of the result."""
    from commpy_amd import _lib
    from commpy_amd.devicelink import DeviceBuf
    from commpy_amd.modulation import QAMModem
    lib = _lib.load()
    md = QAMModem(m)
    nb = md.num_bits_symbol
    rs = np.random.RandomState(m)
    N0 = md.Es / 10 ** (snr_db / 10.0)
    h = md._device_handle()
    guard = 700
    for ns in (1, 2, 63, 64, 65, 127, 255, 256, 257, 1000, 4099, 70001):
        y = md.constellation[rs.randint(0, m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
        want = oracle.demodulate(md.constellation, y, "soft", N0)
        fin = np.isfinite(want)
        d_y = DeviceBuf.from_array(y)
        for mode in (None, "libm", "plain"):
            d_l = DeviceBuf.from_array(np.full(ns * nb + guard, -777.25))
            _lib.demod_set_path(mode)
            try:
                _lib.check(lib.cpx_demod_soft_dev(h, d_y.ptr, ns, float(N0), d_l.ptr, None))
                _lib.check(lib.cpx_stream_sync(None))
                kern = _lib.last_kernel()
            finally:
                _lib.demod_set_path(None)
            got = d_l.to_array((ns * nb + guard,), np.float64)
            d_l.free()
            assert np.all(got[ns * nb:] == -777.25), (ns, mode, kern)
            got = got[:ns * nb]
            assert np.array_equal(np.isfinite(got), fin), (ns, mode, kern)
            assert np.max(np.abs(got[fin] - want[fin]), initial=0.0) < 1e-9, (ns, mode, kern)
            if m >= 64:
                assert (",tab" in kern) == (mode is None) and (",gp" in kern) == (mode != "plain"), kern
        d_y.free()


@pytest.mark.parametrize("kind,m,snr_db", [("psk", 2, 4.0), ("psk", 4, 6.0), ("psk", 8, 9.0), ("psk", 16, 14.0), ("psk", 32, 18.0),
                                           ("custom", 8, 7.0), ("psk", 8, 45.0), ("qam", 64, 12.0)])
def test_generic_soft_demod_fast_kernel(gpu, kind, m, snr_db):
    """Round 6: PSK / arbitrary tables run demod_soft_gen_kernel -- table-driven exp / log, no hypot, the LLRs of a wave stored as one
    contiguous run (any bits per symbol, odd ones too), point-by-point redo of the symbols near the reference's underflow range; a
    caller's output pointer that is only 8-byte aligned is served by the literal kernel (separable constellations too: 64-QAM row).  Against the oracle on every symbol (1e-9; where the oracle is not finite the
    same non-finite value), for every size around the wave / block boundaries, the literal kernel ('libm') beside it, nothing written
    behind the result, and the same values through a misaligned output pointer.  45 dB: most symbols take the redo path."""
    from commpy_amd import _lib
    from commpy_amd.devicelink import DeviceBuf
    from commpy_amd.modulation import Modem, PSKModem, QAMModem
    lib = _lib.load()
    rs = np.random.RandomState(m + int(snr_db))
    md = PSKModem(m) if kind == "psk" else QAMModem(m) if kind == "qam" else Modem(rs.randn(m) + 1j * rs.randn(m))
    nb = md.num_bits_symbol
    N0 = md.Es / 10 ** (snr_db / 10.0)
    h = md._device_handle()
    guard = 300
    for ns in (1, 2, 63, 64, 65, 127, 129, 255, 256, 257, 1000, 4099, 70001):
        y = md.constellation[rs.randint(0, m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
        if ns >= 1000:
            y[5] = 40.0 - 3j                                   # a far outlier: every point probability underflows
        want = oracle.demodulate(md.constellation, y, "soft", N0)
        fin = np.isfinite(want)
        d_y = DeviceBuf.from_array(y)
        for mode, off in ((None, 0), ("libm", 0), (None, 8)):  # off = 8: an output pointer that is only 8-byte aligned
            d_l = DeviceBuf.from_array(np.full(ns * nb + guard + 1, -777.25))
            _lib.demod_set_path(mode)
            try:
                _lib.check(lib.cpx_demod_soft_dev(h, d_y.ptr, ns, float(N0), ctypes.c_void_p(d_l.ptr.value + off), None))
                _lib.check(lib.cpx_stream_sync(None))
                kern = _lib.last_kernel()
            finally:
                _lib.demod_set_path(None)
            got = d_l.to_array((ns * nb + guard + 1,), np.float64)[off // 8:]
            d_l.free()
            fast = "demod_soft_sep_kernel" if kind == "qam" else "demod_soft_gen_kernel"
            assert (fast in kern) == (off == 0 and (mode is None or kind == "qam")), kern
            if off:
                assert kern.startswith("demod_soft_kernel<"), kern
            assert np.all(got[ns * nb:] == -777.25), (ns, mode, off, kern)
            got = got[:ns * nb]
            assert np.array_equal(np.isfinite(got), fin), (ns, mode, off)
            assert np.array_equal(got[~fin], want[~fin], equal_nan=True), (ns, mode, off)
            assert np.max(np.abs(got[fin] - want[fin]), initial=0.0) < 1e-9, (ns, mode, off, np.max(np.abs(got[fin] - want[fin])))
        d_y.free()


# ------------------------------------------------------------------ turbo: the time-major slab at every geometry (round 6)
@pytest.mark.parametrize("tname", ["rsc_legacy_4", "rsc_legacy_8", "k5_23_35"])
def test_turbo_ragged_shapes_vs_oracle(gpu, tname):
    """Since round 6 the slab between the MAP passes is time-major per pair of wavefronts and the interleaver is the row index of
    a pass's loads (csrc/bcjr.hip TurboParams).  Everything that depends on the geometry -- block lengths that are not multiples of
    the 8-step chunk, of the 64-step init tile or of the 256-position final tile; batches that give 1, 2, 4, 8 or 16 codewords per
    pair, with and without a partial last pair; a caller's L_int; 1 to 3 iterations -- against the CPU oracle, every codeword."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import turbo_decode
    tr = make_trellis(tname)
    rs = np.random.RandomState(sum(map(ord, tname)))
    seen = set()
    for N, B, iters, with_lint in ((1, 3, 1, False), (7, 1, 2, False), (8, 2, 1, True), (9, 5, 2, False), (63, 17, 1, False),
                                   (64, 33, 2, True), (65, 4, 3, False), (100, 40, 2, False), (257, 19, 2, True),
                                   (300, 1100, 1, False), (40, 2500, 2, False), (72, 5000, 1, True), (24, 17000, 1, False)):
        perm = rs.permutation(N)
        s, p1, p2 = (np.sign(rs.randn(B, N)) + 0.9 * rs.randn(B, N) for _ in range(3))
        lint = rs.randn(B, N) if with_lint else None
        dec = turbo_decode(s, p1, p2, tr, 0.81, iters, Perm(perm), lint)
        seen.add(_lib.last_kernel().split("wave pairs per workgroup, ")[1].split(" codewords")[0])
        assert dec.shape == (B, N)
        check = range(B) if B <= 64 else sorted(set(rs.randint(0, B, 24)) | {0, B - 1, B - 2})
        for b in check:
            want = oracle.turbo_decode(s[b], p1[b], p2[b], tr, 0.81, iters, Perm(perm), None if lint is None else lint[b])
            assert np.array_equal(dec[b], want), (tname, N, B, iters, b)
    assert len(seen) >= 3, seen                     # several codewords-per-pair geometries were really exercised
