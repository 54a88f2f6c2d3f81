"""Maximum sizes: batches whose tensors pass 2^31 elements / tens of GB, the regime the 288 GB of one MI355X are
meant for.  Inputs are generated on the device (commpy_amd.devicelink stages), so nothing large crosses PCIe.

* Viterbi K=7: 1 050 000 codewords x 2060 LLRs = 2.16e9 float64 (17.3 GB) in ONE call -- noiseless: every decoded
  bit equals its message (also for the codewords stored past the 2^31-th element); Eb/N0 = 3 dB: BER in the band of
  the reference curve (tests/golden/viterbi_ber.npz pins 6.5e-4 at this point).
* LDPC (1944,1296): the WHOLE BASELINE config 4 batch (262 144 codewords) on one GPU, min-sum and sum-product,
  through encode -> 64-QAM -> AWGN -> demod -> BP at 10 dB: every word decoded to what was sent.
"""
import ctypes

import numpy as np
import pytest

from helpers import ldpc_params, make_trellis

pytestmark = pytest.mark.gpu


def _free_bytes():
    try:
        import subprocess
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True, timeout=30).stdout
        rows = [r.split(",") for r in out.strip().splitlines()[1:] if r]
        return int(rows[0][1]) - int(rows[0][2])
    except Exception:
        return None


def _d2h(lib, buf, byte_offset, shape, dtype):
    """Copy a slice of a device buffer: `shape` elements of `dtype` starting `byte_offset` bytes in."""
    from commpy_amd import _lib
    out = np.empty(shape, dtype=dtype)
    _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(out), ctypes.c_void_p(buf.ptr.value + byte_offset), out.nbytes))
    return out


def test_viterbi_two_billion_llrs(gpu):
    from commpy_amd import _lib
    from commpy_amd.devicelink import DeviceBuf
    from commpy_amd.modulation import QAMModem
    free = _free_bytes()
    if free is not None and free < 60e9:
        pytest.skip("needs ~45 GB of free HBM")
    lib = _lib.load()
    tr, md = make_trellis("k7_133_171"), QAMModem(4)
    B, nmsg = 1050000, 1024
    nout, nsym, L, steps = 2060, 1030, 1030, 1035
    assert B * nout > 2 ** 31
    d_msg, d_code = DeviceBuf(B * nmsg), DeviceBuf(B * nout)
    d_sym, d_llr = DeviceBuf(B * nsym * 16), DeviceBuf(B * nout * 8)
    d_dec, d_err = DeviceBuf(B * L), DeviceBuf(B * 4)
    h_tr, h_md = tr._device_handle(), md._device_handle()
    _lib.check(lib.cpx_random_bits_dev(d_msg.ptr, B * nmsg, 5, 0, None))
    _lib.check(lib.cpx_conv_encode_batch_dev(h_tr, d_msg.ptr, B, nmsg, 1, 0, d_code.ptr, nout, None))
    _lib.check(lib.cpx_modulate_dev(h_md, d_code.ptr, B * nsym, d_sym.ptr, None))
    for ebn0, lo, hi in ((None, 0.0, 0.0), (3.0, 5.5e-4, 7.5e-4)):
        if ebn0 is None:
            N0 = 0.5                                                   # clean symbols, LLR = +-8
            _lib.check(lib.cpx_demod_soft_dev(h_md, d_sym.ptr, B * nsym, N0, d_llr.ptr, None))
        else:
            N0 = md.Es / (0.5 * 2 * 10 ** (ebn0 / 10.0))
            sc = float(np.sqrt(N0 / 2))
            _lib.check(lib.cpx_awgn_dev(d_sym.ptr, B * nsym, sc, sc, 11, 1, d_sym.ptr, None))   # in place
            _lib.check(lib.cpx_demod_soft_dev(h_md, d_sym.ptr, B * nsym, float(N0), d_llr.ptr, None))
        _lib.check(lib.cpx_viterbi_decode_batch_dev(h_tr, d_llr.ptr, B, nout, L, steps, 30, 1, d_dec.ptr, None))
        _lib.check(lib.cpx_count_errors_dev(d_msg.ptr, nmsg, d_dec.ptr, L, B, 1, nmsg, d_err.ptr, None))
        _lib.check(lib.cpx_stream_sync(None))
        errs = d_err.to_array((B,), np.int32)
        ber = errs.sum() / float(B * nmsg)
        if ebn0 is None:
            assert errs.max() == 0 and errs.min() == 0
            tail = _d2h(lib, d_dec, (B - 3) * L, (3, L), np.uint8)     # rows past the 2^31-th input element, on the host
            want = _d2h(lib, d_msg, (B - 3) * nmsg, (3, nmsg), np.uint8)
            assert want.any() and not want.all()
            assert np.array_equal(tail[:, :nmsg], want) and not tail[:, nmsg:].any()
        else:
            assert lo <= ber <= hi, ber
            third = B // 3                                           # the error rate is uniform over the batch
            parts = [errs[i * third:(i + 1) * third].sum() / float(third * nmsg) for i in range(3)]
            assert max(parts) < 1.15 * min(parts), parts
    for b in (d_msg, d_code, d_sym, d_llr, d_dec, d_err):
        b.free()
    lib.cpx_release_workspace()


@pytest.mark.parametrize("alg,name", [(1, "MSA"), (0, "SPA")])
def test_ldpc_whole_config4_batch_on_one_gpu(gpu, alg, name):
    from commpy_amd import _lib
    from commpy_amd.channelcoding.ldpc import _device_code
    from commpy_amd.devicelink import DeviceBuf, LdpcEncoder
    from commpy_amd.modulation import QAMModem
    free = _free_bytes()
    if free is not None and free < 90e9:
        pytest.skip("needs ~70 GB of free HBM")
    lib = _lib.load()
    p = ldpc_params("n1944")
    enc, md = LdpcEncoder(p, "gf2"), QAMModem(64)
    B, n, nsym = 262144, 1944, 324
    d_msg, d_code = DeviceBuf(B * enc.k), DeviceBuf(B * n)
    d_sym, d_llr = DeviceBuf(B * nsym * 16), DeviceBuf(B * n * 8)
    d_dec, d_out, d_it = DeviceBuf(B * n), DeviceBuf(B * n * 8), DeviceBuf(B * 4)
    code, h_md = _device_code(p), md._device_handle()
    N0 = 42.0 / ((2.0 / 3) * 6 * 10.0)
    sc = float(np.sqrt(N0 / 2))
    _lib.check(lib.cpx_random_bits_dev(d_msg.ptr, B * enc.k, 30, 0, None))
    enc.encode_dev(d_msg.ptr, B, d_code.ptr)
    _lib.check(lib.cpx_modulate_dev(h_md, d_code.ptr, B * nsym, d_sym.ptr, None))
    _lib.check(lib.cpx_awgn_dev(d_sym.ptr, B * nsym, sc, sc, 31, 1, d_sym.ptr, None))
    _lib.check(lib.cpx_demod_soft_dev(h_md, d_sym.ptr, B * nsym, float(N0), d_llr.ptr, None))
    _lib.check(lib.cpx_scale_f64_dev(d_llr.ptr, B * n, -1.0, d_llr.ptr, None))
    _lib.check(lib.cpx_ldpc_bp_decode_batch_dev(code, d_llr.ptr, B, alg, 50, d_dec.ptr, d_out.ptr, d_it.ptr, None))
    _lib.check(lib.cpx_stream_sync(None))
    its = d_it.to_array((B,), np.int32)
    assert 1 <= its.min() and its.max() < 50 and 4.0 < its.mean() < 6.5, (its.min(), its.max(), its.mean())
    dec = d_dec.to_array((n, B), np.int8)
    sent = d_code.to_array((B, n), np.int8)
    assert np.array_equal(dec.T, sent)
    out = _d2h(lib, d_out, (n - 8) * B * 8, (8, B), np.float64)       # last variable rows: sign agrees with the decision
    assert np.array_equal(out < 0, dec[-8:] == 1)
    for b in (d_msg, d_code, d_sym, d_llr, d_dec, d_out, d_it):
        b.free()
    lib.cpx_release_workspace()


def test_map_and_turbo_long_blocks(gpu):
    """Blocks of 2^20 and 2^22 + 5 steps (the pass addresses its arrays with 31-bit lane offsets into raw buffers, include/commpy_amd.h:
    N < 2^24 for map_decode) and a turbo decode of N = 100 003 (odd length: partial chunk; interleaver through global memory, not LDS)."""
    import oracle
    from helpers import make_trellis
    from commpy_amd.channelcoding import RandInterlv, map_decode, turbo_decode
    tr = make_trellis("rsc_legacy_4")
    rs = np.random.RandomState(1)
    for B, N in ((18, 1 << 20), (3, (1 << 22) + 5)):
        s_ = rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.8
        p_ = rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.8
        L = rs.randn(B, N)
        Le, bits = map_decode(s_, p_, tr, 0.64, L, "decode")
        for b in (0, B - 1):
            Lo, bo = oracle.map_decode(s_[b], p_[b], tr, 0.64, L[b], "decode")
            assert np.max(np.abs(Le[b] - Lo)) < 1e-5, (N, b)
            assert not np.any((bits[b] != bo) & (np.abs(Lo) > 1e-5)), (N, b)
    B, N = 17, 100003
    il = RandInterlv(N, 5)
    r = [rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.9 for _ in range(3)]
    dec = turbo_decode(r[0], r[1], r[2], tr, 0.81, 2, il)
    for b in (0, 16):
        assert np.array_equal(dec[b], oracle.turbo_decode(r[0][b], r[1][b], r[2][b], tr, 0.81, 2, il)), b
