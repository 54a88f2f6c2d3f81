"""The "fp32-fast" precision mode (cpx_set_precision): float32 path metrics in the fused codeword-per-lane Viterbi
kernel.  It is NOT the parity mode -- the contract here is a measured, bounded mismatch rate against the float64 kernels
(convcode.py:661-749 semantics) and an unchanged bit error rate; 'hard' decoding of 0/1 inputs stays bit-identical (its
metrics are small integers).  Everything else must be untouched by the switch."""
import numpy as np
import pytest

import oracle
from helpers import make_trellis

pytestmark = pytest.mark.gpu


@pytest.fixture
def lib():
    from commpy_amd import _lib
    yield _lib
    _lib.set_precision(None)
    _lib.viterbi_set_path(None)


def _config2_batch(B, ebn0_db, seed):
    from commpy_amd.channelcoding import conv_encode_batch
    tr = make_trellis("k7_133_171")
    rs = np.random.RandomState(seed)
    msgs = rs.randint(0, 2, (B, 1024)).astype(np.uint8)
    coded = conv_encode_batch(msgs, tr)
    N0 = 2.0 / (0.5 * 2 * 10 ** (ebn0_db / 10.0))
    y = (2.0 * coded - 1.0) + np.sqrt(N0 / 2) * rs.standard_normal(coded.shape).astype(np.float32)
    return tr, msgs, coded, np.ascontiguousarray((4.0 / N0) * y, dtype=np.float64), y


def test_soft_fast_mode_mismatch_rate_and_ber(gpu, lib):
    import commpy_amd
    from commpy_amd.channelcoding import viterbi_decode
    tr, msgs, _, llr, _ = _config2_batch(65536, 3.0, 10)
    ref = viterbi_decode(llr, tr, None, "soft")
    assert lib.viterbi_last_path() == "fused" and "f32" not in lib.last_kernel()
    commpy_amd.set_precision("fp32-fast")
    assert lib.get_precision() == "fp32-fast"
    fast = viterbi_decode(llr, tr, None, "soft")
    assert "viterbi_cw_fused_kernel" in lib.last_kernel() and ",f32" in lib.last_kernel(), lib.last_kernel()
    commpy_amd.set_precision("fp64-parity")
    again = viterbi_decode(llr, tr, None, "soft")
    assert np.array_equal(again, ref) and "f32" not in lib.last_kernel()          # the switch really switches back
    mism = float(np.mean(fast != ref))
    ber_ref, ber_fast = np.mean(ref[:, :1024] != msgs), np.mean(fast[:, :1024] != msgs)
    print("fp32-fast soft, config-2 batch at 3 dB: mismatching bits vs fp64 %.3e, BER fp64 %.4e, BER fp32 %.4e" % (
        mism, ber_ref, ber_fast))
    assert mism < 2e-5, mism                                                      # measured ~1e-6
    assert abs(ber_fast - ber_ref) < 0.02 * ber_ref + 1e-6


@pytest.mark.parametrize("dtype", ["hard", "unquantized"])
def test_hard_is_identical_and_unquantized_close(gpu, lib, dtype):
    from commpy_amd.channelcoding import viterbi_decode
    tr, msgs, coded, llr, y = _config2_batch(49152, 4.0, 11)
    x = (y > 0).astype(np.float64) if dtype == "hard" else np.ascontiguousarray(y, dtype=np.float64)
    ref = viterbi_decode(x, tr, None, dtype)
    assert lib.viterbi_last_path() == "fused"
    lib.set_precision("fp32-fast")
    fast = viterbi_decode(x, tr, None, dtype)
    assert ",f32" in lib.last_kernel()
    if dtype == "hard":
        assert np.array_equal(fast, ref)                                           # integer metrics: exact in float32
        want = oracle.viterbi_decode_mt(x[:512], tr, None, "hard")
        assert np.array_equal(fast[:512], want)
    else:
        assert np.mean(fast != ref) < 2e-5


def test_fast_mode_leaves_every_other_kernel_alone(gpu, lib):
    """Small Viterbi batches (state-per-lane kernels), the tiled LDPC path, BCJR and the hard-decision demodulator have no float32
    variant: same results, bit for bit."""
    from commpy_amd.channelcoding import ldpc_bp_decode, map_decode, viterbi_decode
    from commpy_amd.modulation import QAMModem
    from helpers import ldpc_params
    tr = make_trellis("k7_133_171")
    tr4 = make_trellis("rsc_legacy_4")
    rs = np.random.RandomState(2)
    x = rs.randn(40, 2 * 200) * 2
    p = ldpc_params("gallager96")
    l = rs.randn(96 * 7) * 3
    md = QAMModem(16)
    y = md.constellation[rs.randint(0, 16, 300)] + 0.2 * (rs.randn(300) + 1j * rs.randn(300))
    s_, p_ = rs.randn(50), rs.randn(50)

    def run():
        lib.ldpc_set_path("tiled")
        try:
            d, o = ldpc_bp_decode(l.copy(), p, "SPA", 5)
        finally:
            lib.ldpc_set_path(None)
        return (viterbi_decode(x, tr, None, "soft"), d, o, md.demodulate(y, "hard"),
                map_decode(s_, p_, tr4, 0.7, np.zeros(50), "compute")[0])

    a = run()
    lib.set_precision("fp32-fast")
    b = run()
    assert "f32" not in lib.last_kernel()
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    with pytest.raises(Exception):
        lib.set_precision("fp16")


@pytest.mark.parametrize("alg", ["MSA", "SPA"])
def test_ldpc_fast_mode_decodes_like_the_parity_mode(gpu, lib, alg):
    """fp32-fast LDPC (float32 state in the LDS-resident kernel): not the 1e-5 LLR contract -- the decoded word of blocks that
    converge, the frame error rate and the iteration counts must stay those of the float64 decoder."""
    from commpy_amd.channelcoding import ldpc_bp_decode
    from helpers import ldpc_params
    p = ldpc_params("n1944")
    n, B = 1944, 4096
    rs = np.random.RandomState(77)
    sigma = 1 / np.sqrt(10 ** 0.3 * (2.0 / 3) * 2)                                  # Eb/N0 = 3 dB, all-zero codeword
    llr = (2.0 * (1.0 + sigma * rs.randn(B * n)) / sigma ** 2)
    d0, o0, i0 = ldpc_bp_decode(llr.copy(), p, alg, 50, return_iterations=True)
    assert "ldpc_resident_kernel" in lib.last_kernel()
    lib.set_precision("fp32-fast")
    d1, o1, i1 = ldpc_bp_decode(llr.copy(), p, alg, 50, return_iterations=True)
    assert "ldpc_resident_f32_kernel" in lib.last_kernel(), lib.last_kernel()
    lib.set_precision(None)
    ok0, ok1 = ~d0.any(axis=0), ~d1.any(axis=0)                                     # decoded to the transmitted (all-zero) word
    fer0, fer1 = 1 - ok0.mean(), 1 - ok1.mean()
    both = (i0 < 50) & (i1 < 50)
    same_word = np.all(d0[:, both] == d1[:, both], axis=0).mean()
    print("fp32-fast LDPC %s at 3 dB: FER fp64 %.4f, fp32 %.4f; mean iterations %.2f / %.2f; identical dec_word on %.5f of the "
          "blocks both decoders converge on; sign(out_llrs) agreement %.6f" % (
              alg, fer0, fer1, i0.mean(), i1.mean(), same_word, np.mean(np.signbit(o0[:, both]) == np.signbit(o1[:, both]))))
    assert same_word > 0.999
    assert abs(fer1 - fer0) <= 0.01 + 0.2 * fer0
    assert abs(i1.mean() - i0.mean()) < 0.5


def test_precision_context_manager_restores_the_mode(gpu, lib):
    """``with commpy_amd.precision('fp32-fast')`` switches the mode for the block only -- also when the block raises."""
    import commpy_amd
    assert lib.get_precision() == "fp64-parity"
    with commpy_amd.precision("fp32-fast"):
        assert lib.get_precision() == "fp32-fast"
        with commpy_amd.precision("fp64-parity"):
            assert lib.get_precision() == "fp64-parity"
        assert lib.get_precision() == "fp32-fast"
    assert lib.get_precision() == "fp64-parity"
    with pytest.raises(RuntimeError):
        with commpy_amd.precision("fp32-fast"):
            raise RuntimeError("boom")
    assert lib.get_precision() == "fp64-parity"


# ---- fp32-fast soft demodulation (round 4): float32 log-sum-exp kernels, float64 arrays in and out -------------------------------------
def _modem(kind, M):
    from commpy_amd.modulation import PSKModem, QAMModem
    return QAMModem(M) if kind == "qam" else PSKModem(M)


@pytest.mark.parametrize("kind,M", [("qam", 4), ("qam", 16), ("qam", 64), ("qam", 256), ("psk", 2), ("psk", 4), ("psk", 8), ("psk", 16)])
def test_soft_demod_fast_mode_tolerance(gpu, lib, kind, M):
    """Contract of the float32 demodulator: |LLR_fast - LLR_fp64| <= 2e-5 + 4e-6 |LLR_fp64| (modulation.py:125-137 in float64 is the
    reference), same hard decisions wherever the float64 LLR is not within 1e-3 of zero; ragged sizes keep their layout."""
    md = _modem(kind, M)
    rs = np.random.RandomState(M + len(kind))
    Es = float(np.mean(np.abs(md.constellation) ** 2))
    worst_abs, worst_rel = 0.0, 0.0
    for snr_db, Ns in ((0.0, 4099), (10.0, 65), (20.0, 40000), (28.0, 1), (14.0, 63), (27.0, 3000)):
        N0 = Es / 10 ** (snr_db / 10)
        y = md.constellation[rs.randint(0, M, Ns)] + np.sqrt(N0 / 2) * (rs.randn(Ns) + 1j * rs.randn(Ns))
        lib.set_precision(None)
        ref = md.demodulate(y, "soft", N0)
        assert "f32" not in lib.last_kernel()
        lib.set_precision("fp32-fast")
        fast = md.demodulate(y, "soft", N0)
        assert "_f32_kernel" in lib.last_kernel(), lib.last_kernel()
        assert fast.shape == ref.shape and fast.dtype == np.float64
        fin = np.isfinite(ref) & (np.abs(ref) < 600)                # beyond: the float64 sums are among the denormals themselves
        assert np.all(np.isfinite(fast))
        err = np.abs(fast - ref)[fin]
        worst_abs = max(worst_abs, float(err.max(initial=0.0)))
        worst_rel = max(worst_rel, float((err / (2e-5 + 4e-6 * np.abs(ref[fin]))).max(initial=0.0)))
        assert np.all(err <= 2e-5 + 4e-6 * np.abs(ref[fin])), (snr_db, float(err.max()))
        sure = fin & (np.abs(ref) > 1e-3)
        assert np.array_equal(fast[sure] > 0, ref[sure] > 0)
    print("fp32-fast demod %s-%d: worst |dLLR| %.2e, worst fraction of the allowance %.2f" % (kind, M, worst_abs, worst_rel))


def test_soft_demod_fast_mode_is_finite_where_float64_underflows(gpu, lib):
    """At Es/N0 far beyond any operating point the reference's sums underflow (-inf / NaN, modulation.py:134-137) and fp64-parity
    reproduces that; the log-sum-exp form does not underflow: finite LLRs with the sign of the nearest point's bits."""
    from commpy_amd.modulation import QAMModem
    md = QAMModem(64)
    rs = np.random.RandomState(5)
    idx = rs.randint(0, 64, 5000)
    y = md.constellation[idx] + 0.01 * (rs.randn(5000) + 1j * rs.randn(5000))
    N0 = 1e-3
    ref = md.demodulate(y, "soft", N0)
    assert not np.all(np.isfinite(ref))
    with __import__("commpy_amd").precision("fp32-fast"):
        fast = md.demodulate(y, "soft", N0)
        hard = md.demodulate(y, "hard")
    assert np.all(np.isfinite(fast))
    assert np.array_equal((fast > 0).astype(np.int8), hard.astype(np.int8))
    fin = np.isfinite(ref) & (np.abs(ref) < 600)
    assert np.all(np.abs(fast - ref)[fin] <= 2e-5 + 4e-6 * np.abs(ref[fin]))


def test_soft_demod_fast_mode_scaled_and_device_entry(gpu, lib):
    """cpx_demod_soft_scaled_dev (the LDPC sign convention, scale = -1) in fp32-fast: exactly the negated LLRs, same layout; a noise
    variance outside the float32 range keeps the float64 kernel."""
    from commpy_amd.devicelink import DeviceBuf
    from commpy_amd.modulation import PSKModem, QAMModem
    so = lib.load()
    rs = np.random.RandomState(8)
    for md in (QAMModem(16), PSKModem(8)):
        ns, nb = 777, md.num_bits_symbol
        y = md.constellation[rs.randint(0, md.m, ns)] + 0.3 * (rs.randn(ns) + 1j * rs.randn(ns))
        ref = md.demodulate(y, "soft", 0.4)
        lib.set_precision("fp32-fast")
        d_y = DeviceBuf(y.nbytes)
        lib.check(so.cpx_memcpy_h2d(d_y.ptr, lib.ptr(y), y.nbytes))
        outs = []
        for scale in (1.0, -1.0):
            d_l = DeviceBuf(ns * nb * 8)
            lib.check(so.cpx_demod_soft_scaled_dev(md._device_handle(), d_y.ptr, ns, 0.4, scale, d_l.ptr, None))
            assert "_f32_kernel" in lib.last_kernel()
            o = np.empty(ns * nb)
            lib.check(so.cpx_memcpy_d2h(lib.ptr(o), d_l.ptr, o.nbytes))
            outs.append(o)
        assert np.array_equal(outs[1], -outs[0])
        assert np.all(np.abs(outs[0] - ref) <= 2e-5 + 4e-6 * np.abs(ref))
        md.demodulate(y, "soft", 1e-40)
        assert "f32" not in lib.last_kernel()
        lib.set_precision(None)


@pytest.mark.parametrize("states,N,iters", [(4, 1024, 6), (8, 512, 4), (4, 203, 2)])
def test_turbo_fast_mode_float32_slab(gpu, lib, states, N, iters):
    """Round 5: `fp32-fast` turbo decoding keeps the float64 arithmetic of the MAP passes and stores the slab BETWEEN them in float32
    (channel factors, the smaller of (p0, p1) with a sign, the LLRs of the last two passes): half the bytes per pass and interleave
    stage.  Not the parity mode: the contract is an unchanged bit error rate and a bounded fraction of bits that differ from the float64
    decode at the code's operating point (measured ~1e-6: only decisions whose LLR sits within float32 rounding of zero can move)."""
    import commpy_amd
    from commpy_amd.channelcoding import RandInterlv, turbo_decode
    from commpy_amd.devicelink import turbo_encode_gpu
    tr = make_trellis("rsc_legacy_4" if states == 4 else "rsc_legacy_8")
    rs = np.random.RandomState(states + N)
    B = 4096 if N >= 512 else 1500
    il = RandInterlv(N, 99)
    msgs = rs.randint(0, 2, (B, N))
    nv = 1 / (2 * (1.0 / 3) * 10 ** (1.5 / 10.0))
    s, p1, p2 = (a[:, :N] * 2.0 - 1 + np.sqrt(nv) * rs.standard_normal((B, N)) for a in turbo_encode_gpu(msgs, tr, tr, il))
    Lint = rs.randn(B, N) * 0.5
    for use_L in (False, True):
        ref = turbo_decode(s, p1, p2, tr, nv, iters, il, Lint if use_L else None)
        assert "f32" not in lib.last_kernel(), lib.last_kernel()
        commpy_amd.set_precision("fp32-fast")
        try:
            fast = turbo_decode(s, p1, p2, tr, nv, iters, il, Lint if use_L else None)
            assert "f32 slab" in lib.last_kernel(), lib.last_kernel()
            again = turbo_decode(s, p1, p2, tr, nv, iters, il, Lint if use_L else None)
        finally:
            commpy_amd.set_precision("fp64-parity")
        assert np.array_equal(fast, again)                                           # deterministic
        assert np.array_equal(turbo_decode(s, p1, p2, tr, nv, iters, il, Lint if use_L else None), ref)   # the switch switches back
        mism = float(np.mean(fast != ref))
        ber_ref, ber_fast = float(np.mean(ref != msgs)), float(np.mean(fast != msgs))
        print("fp32-fast turbo %d states N=%d %d its L_int=%s: bits differing from fp64 %.3e, BER fp64 %.4e, fp32 slab %.4e" % (
            states, N, iters, use_L, mism, ber_ref, ber_fast))
        assert mism < 1e-4, mism
        assert abs(ber_fast - ber_ref) <= 0.05 * ber_ref + 2e-5, (ber_ref, ber_fast)
    # extreme priors and a clean high-SNR batch: finite arithmetic, no crash, decisions follow the priors / the channel
    big = np.where(msgs == 1, 60.0, -60.0)
    commpy_amd.set_precision("fp32-fast")
    try:
        assert np.mean(turbo_decode(s, p1, p2, tr, nv, iters, il, big) != msgs) < 1e-3
    finally:
        commpy_amd.set_precision("fp64-parity")
