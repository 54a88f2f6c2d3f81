"""Randomised parity: random convolutional codes (feed-forward, recursive-systematic, k = 2), random lengths,
traceback depths and metric types -- HIP Viterbi / MAP vs the CPU oracle, bit-exact / 1e-5."""
import warnings

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _random_trellis(rs):
    from commpy_amd.channelcoding import Trellis
    kind = rs.randint(0, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if kind == 0:                                   # feed-forward, k = 1, memory 1..6, n = 2..3
            m = int(rs.randint(1, 7))
            n = int(rs.randint(2, 4))
            g = rs.randint(1, 2 ** (m + 1), (1, n))
            g[0, 0] |= 1 | (1 << m)                     # make the code use its full memory
            return Trellis(np.array([m]), g)
        if kind == 1:                                   # recursive systematic, matrix feedback
            m = int(rs.randint(1, 5))
            fb = int(rs.randint(1, 2 ** (m + 1))) | 1 | (1 << m)
            g = np.array([[1 << 0, int(rs.randint(1, 2 ** (m + 1))) | 1]])
            return Trellis(np.array([m]), g, np.array([[fb]]), 'rsc')
        m1, m2 = int(rs.randint(1, 3)), int(rs.randint(1, 3))   # k = 2, n = 3
        g = rs.randint(0, 2 ** (max(m1, m2) + 1), (2, 3))
        g[0, 0] |= 1
        g[1, 2] |= 1
        return Trellis(np.array([m1, m2]), g)


@pytest.mark.parametrize("seed", range(12))
def test_viterbi_random_codes(gpu, seed):
    from commpy_amd.channelcoding import conv_encode_batch, viterbi_decode
    rs = np.random.RandomState(1000 + seed)
    for _ in range(6):
        tr = _random_trellis(rs)
        if tr.number_states > 128:
            continue
        try:
            tr._device_handle()
        except ValueError:                              # irregular trellis (parallel branches overflow in-degree): reference breaks too
            continue
        nbits = int(rs.randint(8, 200))
        nbits -= nbits % tr.k
        nbits = max(nbits, 4 * tr.k)
        B = int(rs.randint(1, 70))
        term = 'term' if rs.rand() < 0.5 else 'cont'
        coded = conv_encode_batch(rs.randint(0, 2, (B, nbits)), tr, term).astype(float)
        dtype = ('hard', 'soft', 'unquantized')[rs.randint(0, 3)]
        if dtype == 'hard':
            rx = np.where(rs.rand(*coded.shape) < 0.08, 1 - coded, coded)
        elif dtype == 'soft':
            rx = (4.0 * coded - 2) + rs.randn(*coded.shape) * 2.5
        else:
            rx = (2.0 * coded - 1) + rs.randn(*coded.shape)
        L = int(rx.shape[1] * tr.k / tr.n)
        steps = int((L + tr.total_memory) / tr.k) - 1
        tb = None if rs.rand() < 0.4 else int(rs.randint(2, max(3, min(60, steps + 1))))
        got = viterbi_decode(rx, tr, tb, dtype)
        want = oracle.viterbi_decode(rx, tr, tb, dtype)
        assert np.array_equal(got, want), (seed, tr.k, tr.n, tr.number_states, nbits, B, term, dtype, tb)


@pytest.mark.parametrize("seed", range(4))
def test_map_random_rsc_codes(gpu, seed):
    from commpy_amd.channelcoding import Trellis, map_decode
    rs = np.random.RandomState(2000 + seed)
    for _ in range(4):
        m = int(rs.randint(1, 5))
        fb = int(rs.randint(1, 2 ** (m + 1))) | 1 | (1 << m)
        g = np.array([[1, int(rs.randint(1, 2 ** (m + 1))) | 1]])
        tr = Trellis(np.array([m]), g, np.array([[fb]]), 'rsc')
        N, B = int(rs.randint(5, 150)), int(rs.randint(1, 40))
        s, p, li = rs.randn(B, N) * 1.3, rs.randn(B, N) * 1.3, rs.randn(B, N)
        nv = float(rs.uniform(0.3, 2.0))
        L, bits = map_decode(s, p, tr, nv, li, 'decode')
        for b in range(min(B, 3)):
            Lo, bo = oracle.map_decode(s[b], p[b], tr, nv, li[b], 'decode')
            assert np.max(np.abs(L[b] - Lo)) < 1e-5, (seed, m, N)
            assert not np.any((bits[b] != bo) & (np.abs(Lo) > 1e-5))
