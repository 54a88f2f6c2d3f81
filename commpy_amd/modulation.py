"""PSK / QAM modems with MI355X demodulation.

Same public names, arguments and return conventions as /root/reference/commpy/modulation.py:39-262
(``Modem``, ``PSKModem``, ``QAMModem``).  Construction, Gray re-indexing, ``modulate`` and ``Es`` stay on
the host; ``demodulate`` ('hard' and 'soft') runs on the GPU through ``cpx_demod_hard`` /
``cpx_demod_soft`` (csrc/demod.hip).  MIMO detectors and OFDM helpers of the reference module are
out of scope.
"""
import ctypes

import numpy as np

from commpy_amd import _lib
from commpy_amd.utilities import signal_power

__all__ = ['PSKModem', 'QAMModem', 'Modem']


def _gray_rank(m):
    """Position of each label in the reflected Gray sequence: the reference re-indexes the
    constellation with ``constellation[gray_sequence.argsort()]`` (modulation.py:71-75), where
    ``gray_sequence[i] = i ^ (i >> 1)`` (SymPy's GrayCode order)."""
    idx = np.arange(m)
    return (idx ^ (idx >> 1)).argsort()


class Modem:
    """Custom modem -- modulation.py:39-172.  ``constellation`` must have a power-of-two length."""

    def __init__(self, constellation, reorder_as_gray=True):
        self._cpx_handles = None
        if reorder_as_gray:
            self.constellation = np.array(constellation)[_gray_rank(len(constellation))]
        else:
            self.constellation = constellation

    @property
    def constellation(self):
        return self._constellation

    @constellation.setter
    def constellation(self, value):
        num_bits_symbol = np.log2(len(value))
        if num_bits_symbol != int(num_bits_symbol):
            raise ValueError('Constellation length must be a power of 2.')
        self._constellation = np.array(value)
        self.Es = signal_power(self.constellation)
        self.m = self._constellation.size
        self.num_bits_symbol = int(num_bits_symbol)
        self._drop_handle()

    def modulate(self, input_bits):
        """Bits -> symbols (host): label = MSB-first value of each group of ``num_bits_symbol`` bits
        (modulation.py:79-98)."""
        bits = np.asarray(input_bits).astype(np.int64)
        nb = self.num_bits_symbol
        groups = bits.reshape(-1, nb)
        labels = groups.dot(1 << np.arange(nb - 1, -1, -1))
        return self._constellation[labels]

    def demodulate(self, input_symbols, demod_type, noise_var=0):
        """Symbols -> bits/LLRs on MI355X; same signature/return as modulation.py:100.

        'hard': int8 bits of the nearest point (first minimum), MSB first.
        'soft': float64 LLRs ``log P(1)/P(0)`` with ``noise_var`` used as-is (no factor 2).
        """
        if demod_type not in ('hard', 'soft'):
            raise ValueError('demod_type must be "hard" or "soft"')
        lib = _lib.load()
        y = np.ascontiguousarray(np.atleast_1d(input_symbols), dtype=np.complex128).reshape(-1)
        ns = y.size
        h = self._device_handle()
        if demod_type == 'hard':
            out = np.zeros(ns * self.num_bits_symbol, dtype=np.int8)
            if ns:
                _lib.check(lib.cpx_demod_hard(h, _lib.ptr(y), ns, _lib.ptr(out)))
            return out
        out = np.zeros(ns * self.num_bits_symbol)
        if ns:
            _lib.check(lib.cpx_demod_soft(h, _lib.ptr(y), ns, float(noise_var), _lib.ptr(out)))
        return out

    # -- device handle -----------------------------------------------------------------------
    def _device_handle(self):
        """Opaque cpx_modem* of the current device (created on first use, one per device)."""
        if self._cpx_handles is None:
            def create():
                c = np.ascontiguousarray(self._constellation, dtype=np.complex128)
                h = ctypes.c_void_p()
                _lib.check(_lib.load().cpx_modem_create(_lib.ptr(c), int(self.m), ctypes.byref(h)))
                return h
            self._cpx_handles = _lib.DeviceHandles(create, 'cpx_modem_destroy')
        return self._cpx_handles.get()

    def _drop_handle(self):
        hs = getattr(self, '_cpx_handles', None)
        if hs is not None:
            hs.drop()
        self._cpx_handles = None

    def demodulate_viterbi_hard(self, input_symbols, trellis, tb_depth=None):
        """``viterbi_decode(self.demodulate(y, 'hard'), trellis, tb_depth, 'hard')`` in ONE kernel
        (modulation.py:121-123 feeding convcode.py:578-580, 661-749): the hard decisions are taken inside the
        Viterbi kernel while it prepares the branch metrics, the int8 bits never exist in HBM.  ``input_symbols``:
        1-D (one codeword) or ``[B, nsym]``; returns what the two calls return (int64, tail included).
        Trellises above 64 states take the two calls (the fused kernel keeps one state per lane)."""
        from commpy_amd.channelcoding.convcode import _viterbi_sizes, viterbi_decode
        y = np.ascontiguousarray(input_symbols, dtype=np.complex128)
        single = y.ndim == 1
        y2 = np.atleast_2d(y)
        B, nsym = y2.shape
        length = nsym * self.num_bits_symbol
        if trellis.number_states > 64 or length == 0:
            bits = self.demodulate(y2.reshape(-1), 'hard').reshape(B, length)
            return viterbi_decode(bits[0] if single else bits, trellis, tb_depth, 'hard')
        L, T, tb = _viterbi_sizes(length, trellis, tb_depth)
        out = np.zeros((B, L), dtype=np.uint8)
        if B and L:
            _lib.check(_lib.load().cpx_demod_hard_viterbi_batch(self._device_handle(), trellis._device_handle(),
                                                                _lib.ptr(y2), B, nsym, L, T, tb, _lib.ptr(out)))
        out = out.astype(np.int64)
        return out[0] if single else out


class PSKModem(Modem):
    """m-PSK: ``exp(1j * arange(0, 2*pi, 2*pi/m))`` Gray re-indexed -- modulation.py:175-210."""

    def __init__(self, m):
        num_bits_symbol = np.log2(m)
        if num_bits_symbol != int(num_bits_symbol):
            raise ValueError('Constellation length must be a power of 2.')
        super().__init__(np.exp(1j * np.arange(0, 2 * np.pi, 2 * np.pi / m)))


class QAMModem(Modem):
    """Square m-QAM on the odd-integer grid, snake ordered then Gray re-indexed -- modulation.py:213-262."""

    def __init__(self, m):
        side = np.sqrt(m)
        if side != int(side):
            raise ValueError('m must lead to a square QAM.')
        side = int(side)
        pam = np.arange(-side + 1, side, 2)
        # column c (real part pam[c]) runs upwards for even c, downwards for odd c
        imag = np.tile(np.hstack((pam, pam[::-1])), side // 2)
        real = pam.repeat(side)
        super().__init__(imag * 1j + real)
