"""BASELINE config 2 acceptance: "BER-vs-Eb/N0 curves overlapping CommPy's".  The reference points in
tests/golden/viterbi_ber.npz are error counts of the live reference decoder (K=7 (133,171) soft Viterbi, 1024-bit
blocks, QPSK + AWGN, make_golden.gen_viterbi_ber; round 6: 224 codewords = 229 376 bits per point, with the error count of every
codeword).  The GPU curve uses its own noise stream and 18x more bits, so the comparison is statistical: a two-sample test on the
errors per codeword with the spread measured on both samples."""
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
    for i, (e, ref_cw) in enumerate(zip(g["ebn0"], g["cw_errors"])):
        rs = np.random.RandomState(7000 + i)
        msg = rs.randint(0, 2, (B, L))
        coded = conv_encode_batch(msg, tr)
        assert coded.shape == (B, 2060)
        sym = md.modulate(coded.reshape(-1))
        N0 = md.Es / (0.5 * 2 * 10 ** (e / 10.0))                      # rate 1/2, 2 bits per symbol
        y = sym + np.sqrt(N0 / 2) * (rs.randn(sym.size) + 1j * rs.randn(sym.size))
        llr = md.demodulate(y, "soft", N0).reshape(B, -1)
        dec = viterbi_decode(llr, tr, None, "soft")
        cw = (dec[:, :L] != msg).sum(1).astype(float)                  # errors per codeword: the burstiness is MEASURED on both sides
        curve.append(float(cw.mean() / L))
        ref = ref_cw.astype(float)
        se = np.sqrt(cw.var(ddof=1) / cw.size + ref.var(ddof=1) / ref.size)
        # two-sample test on the mean errors per codeword (224 reference codewords, 4096 here): 4.5 standard errors, and 0.05 errors per
        # codeword of slack for the 4 dB point where the reference saw 5 errors in 229 376 bits
        assert abs(cw.mean() - ref.mean()) <= 4.5 * se + 0.05, (float(e), cw.mean(), ref.mean(), se)
    assert all(a > b for a, b in zip(curve, curve[1:])), curve         # strictly falling waterfall
    assert curve[-1] < 1e-4 < curve[0]
