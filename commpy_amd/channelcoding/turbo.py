"""Turbo codes: host encoder + MI355X BCJR/MAP and fused turbo decoder.

Same public names, arguments and return conventions as /root/reference/commpy/channelcoding/turbo.py:

* ``turbo_encode`` (turbo.py:14-59)   host (input generator; keeps the reference's quirks B3/B4);
* ``map_decode``   (turbo.py:163-251) DEVICE -> ``cpx_map_decode_batch``  (csrc/bcjr.hip);
* ``turbo_decode`` (turbo.py:254-333) DEVICE -> ``cpx_turbo_decode_batch`` (one launch per MAP pass + small interleaver launches; flagged codewords redone exactly).

2-D inputs ``[B, N]`` decode a batch of independent codewords (extension).
"""
import numpy as np

from commpy_amd import _lib
from commpy_amd.channelcoding.convcode import conv_encode

__all__ = ['turbo_encode', 'map_decode', 'turbo_decode']


def turbo_encode(msg_bits, trellis1, trellis2, interleaver):
    """Parallel-concatenated rate-1/3 turbo encoder (host) -- turbo.py:14-59.

    Returns ``[sys_stream, non_sys_stream_1, non_sys_stream_2]`` exactly like the reference,
    including: ``'rsc'`` passed as the *termination* argument (turbo.py:47: ``conv_encode`` then reserves room
    for a tail -- anything but 'cont' does, convcode.py:509-512 -- but clocks none, which only 'term' would,
    :538, so the reserved tail stays zero), the tailed systematic stream being interleaved, and the second parity stream
    keeping ``conv_encode``'s unpunctured length with a zero tail (quirks B3/B4) -- use
    ``non_sys_stream_2[:N]``.
    """
    stream = conv_encode(msg_bits, trellis1, 'rsc')
    sys_stream = stream[::2]
    non_sys_stream_1 = stream[1::2]
    interlv_msg_bits = interleaver.interlv(sys_stream)
    puncture_matrix = np.array([[0, 1]])
    non_sys_stream_2 = conv_encode(interlv_msg_bits, trellis2, 'rsc', puncture_matrix)
    sys_stream = sys_stream[0:-trellis1.total_memory]
    non_sys_stream_1 = non_sys_stream_1[0:-trellis1.total_memory]
    non_sys_stream_2 = non_sys_stream_2[0:-trellis2.total_memory]
    return [sys_stream, non_sys_stream_1, non_sys_stream_2]


def _batch(a):
    a = np.asarray(a)
    single = a.ndim == 1
    return _lib.as_f64(a.reshape(1, -1) if single else a), single


def map_decode(sys_symbols, non_sys_symbols, trellis, noise_variance, L_int, mode='decode'):
    """MAP (BCJR) decoder on MI355X; same signature/return as turbo.py:163.

    Returns the list ``[L_ext, decoded_bits]``: ``L_ext = L_int + log(app1/app0)`` (float64) and the
    hard decisions ``L_ext > 0`` as int64 (all zeros in 'compute' mode, like the reference).
    """
    lib = _lib.load()
    s, single = _batch(sys_symbols)
    p, _ = _batch(non_sys_symbols)
    li, _ = _batch(L_int)
    if not (s.shape == p.shape == li.shape):
        raise ValueError('sys_symbols, non_sys_symbols and L_int must have the same shape')
    B, N = s.shape
    L_ext = np.zeros((B, N))
    bits = np.zeros((B, N), dtype=np.uint8)
    if B and N:
        _lib.check(lib.cpx_map_decode_batch(trellis._device_handle(), _lib.ptr(s), _lib.ptr(p), _lib.ptr(li), B, N,
                                            float(noise_variance), 1 if mode == 'decode' else 0, _lib.ptr(L_ext),
                                            _lib.ptr(bits)))
    bits = bits.astype(np.int64)
    return [L_ext[0], bits[0]] if single else [L_ext, bits]


def turbo_decode(sys_symbols, non_sys_symbols_1, non_sys_symbols_2, trellis, noise_variance, number_iterations,
                 interleaver, L_int=None):
    """Turbo decoder on MI355X; same signature/return as turbo.py:254 (int64 decoded bits).

    ``interleaver`` is duck-typed like the reference but must expose its permutation as ``p_array``
    (``interlv: out = in[p]``), which becomes a kernel input.
    """
    lib = _lib.load()
    s, single = _batch(sys_symbols)
    p1, _ = _batch(non_sys_symbols_1)
    p2, _ = _batch(non_sys_symbols_2)
    if not (s.shape == p1.shape == p2.shape):
        raise ValueError('the three symbol streams must have the same shape')
    B, N = s.shape
    perm = _lib.as_i32(interleaver.p_array)
    if perm.shape != (N,):
        raise ValueError('interleaver length does not match the block length')
    li = None
    if L_int is not None:
        li, _ = _batch(L_int)
        if li.shape != s.shape:
            raise ValueError('L_int must have the shape of sys_symbols')
    bits = np.zeros((B, N), dtype=np.uint8)
    if B and N:
        _lib.check(lib.cpx_turbo_decode_batch(trellis._device_handle(), _lib.ptr(s), _lib.ptr(p1), _lib.ptr(p2),
                                              None if li is None else _lib.ptr(li), _lib.ptr(perm), B, N,
                                              float(noise_variance), int(number_iterations), _lib.ptr(bits)))
    bits = bits.astype(np.int64)
    return bits[0] if single else bits
