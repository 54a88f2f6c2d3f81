"""BASELINE config 2 acceptance: "BER-vs-Eb/N0 curves overlapping CommPy's".  The reference points in
tests/golden/viterbi_ber.npz are error counts of the live reference decoder (K=7 (133,171) soft Viterbi, 1024-bit
blocks, QPSK + AWGN, make_golden.gen_viterbi_ber).  The GPU curve uses its own noise stream and 200x more bits, so
the comparison is statistical: the reference count has to be plausible under the GPU's (tighter) BER estimate."""
import numpy as np
import pytest

from helpers import golden

pytestmark = pytest.mark.gpu


def test_config2_ber_curve_overlays_reference(gpu):
    from commpy_amd.channelcoding import Trellis, conv_encode_batch, viterbi_decode
    from commpy_amd.modulation import QAMModem
    g = golden("viterbi_ber")
    tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
    md = QAMModem(4)
    B, L = 4096, 1024
    curve = []
    for i, (e, ref_err, ref_bits) in enumerate(zip(g["ebn0"], g["errors"], g["bits"])):
        rs = np.random.RandomState(7000 + i)
        msg = rs.randint(0, 2, (B, L))
        coded = conv_encode_batch(msg, tr)
        assert coded.shape == (B, 2060)
        sym = md.modulate(coded.reshape(-1))
        N0 = md.Es / (0.5 * 2 * 10 ** (e / 10.0))                      # rate 1/2, 2 bits per symbol
        y = sym + np.sqrt(N0 / 2) * (rs.randn(sym.size) + 1j * rs.randn(sym.size))
        llr = md.demodulate(y, "soft", N0).reshape(B, -1)
        dec = viterbi_decode(llr, tr, None, "soft")
        ber = float(np.mean(dec[:, :L] != msg))
        curve.append(ber)
        expect = ber * ref_bits                                        # errors the reference run should have seen
        # Viterbi error events are bursts of ~5-10 bits: inflate the binomial width accordingly
        sigma = np.sqrt(8.0 * max(expect, 1.0))
        assert abs(ref_err - expect) <= 4 * sigma + 3, (float(e), ber, int(ref_err), expect)
    assert all(a > b for a, b in zip(curve, curve[1:])), curve         # strictly falling waterfall
    assert curve[-1] < 1e-4 < curve[0]
