"""Turbo codes: host encoder + MI355X BCJR/MAP and fused turbo decoder.

Same public names, arguments and return conventions as /root/reference/commpy/channelcoding/turbo.py:

* ``turbo_encode`` (turbo.py:14-59)   host (input generator; keeps the reference's quirks B3/B4);
* ``map_decode``   (turbo.py:163-251) DEVICE -> ``cpx_map_decode_batch``  (csrc/bcjr.hip);
* ``turbo_decode`` (turbo.py:254-333) DEVICE -> ``cpx_turbo_decode_batch`` (one launch per MAP pass + small interleaver launches; flagged codewords redone exactly).

2-D inputs ``[B, N]`` decode a batch of independent codewords (extension).
"""
import numpy as np

from commpy_amd import _lib
from commpy_amd.channelcoding.convcode import conv_encode_batch, device_trellis

__all__ = ['turbo_encode', 'map_decode', 'turbo_decode']


def turbo_encode(msg_bits, trellis1, trellis2, interleaver):
    """Parallel-concatenated rate-1/3 turbo encoder (host input generator) -- what turbo.py:14-59 returns, also for a
    batch ``[B, N]`` of messages (extension; lists of ``[B, ...]`` arrays then).

    Returns ``[systematic, parity_1, parity_2]``.  Two properties of the reference's result are kept on purpose:

    * no component encoder is terminated.  The reference passes ``'rsc'`` where ``conv_encode`` expects the termination mode
      (turbo.py:47,53): room for a tail is reserved (anything but 'cont' does that, convcode.py:509-512) but never clocked
      (only 'term' would, :538), and the reserved part is cut off again -- so ``systematic`` is the message and ``parity_1``
      the parity of the plain walk from state 0;
    * ``parity_2`` has ``2 (N + m2) - m2`` entries: the N parity bits of the permuted message first, zeros behind them
      (the punctured stream keeps the unpunctured length, convcode.py:552-556) -- use ``parity_2[:N]``.

    The interleaver sees the systematic stream with its all-zero reserved tail, like in the reference; one that exposes
    ``p_array`` (all of commpy's do, interleavers.py:13-47) is applied as one gather for the whole batch.
    """
    msgs = np.asarray(msg_bits)
    single = msgs.ndim == 1
    rows = msgs.reshape(1, -1) if single else msgs
    B, N = rows.shape
    m1, m2 = int(trellis1.total_memory), int(trellis2.total_memory)
    pairs1 = conv_encode_batch(rows, trellis1, 'rsc')                    # [B, 2 (N + m1)]: (systematic, parity), zero tail
    tailed = pairs1[:, 0::2]                                             # [B, N + m1]
    perm = getattr(interleaver, 'p_array', None)
    if perm is not None:
        permuted = tailed[:, np.asarray(perm)]
    else:                                                                # duck-typed interleaver: only .interlv
        permuted = np.stack([np.asarray(interleaver.interlv(r)) for r in tailed])
    pairs2 = conv_encode_batch(permuted, trellis2, 'rsc')
    n2 = pairs2.shape[1] // 2                                            # len(permuted) + m2 parity positions, tail zero
    parity_2 = np.zeros((B, 2 * n2 - m2), dtype=pairs2.dtype)
    parity_2[:, :n2] = pairs2[:, 1::2]
    out = [tailed[:, :tailed.shape[1] - m1], pairs1[:, 1::2][:, :tailed.shape[1] - m1], parity_2]
    return [o[0] for o in out] if single else out


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
        _lib.check(lib.cpx_map_decode_batch(device_trellis(trellis), _lib.ptr(s), _lib.ptr(p), _lib.ptr(li), B, N,
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
        _lib.check(lib.cpx_turbo_decode_batch(device_trellis(trellis), _lib.ptr(s), _lib.ptr(p1), _lib.ptr(p2),
                                              None if li is None else _lib.ptr(li), _lib.ptr(perm), B, N,
                                              float(noise_variance), int(number_iterations), _lib.ptr(bits)))
    bits = bits.astype(np.int64)
    return bits[0] if single else bits
