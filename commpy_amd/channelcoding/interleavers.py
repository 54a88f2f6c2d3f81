"""Interleavers (host) -- same names as /root/reference/commpy/channelcoding/interleavers.py.

The permutation array ``p_array`` is a kernel input of the fused turbo decoder
(``cpx_turbo_decode_batch``); ``interlv`` is a gather ``out = in[p]`` (interleavers.py:13-28) and
``deinterlv`` the matching scatter ``out[p[i]] = in[i]`` (interleavers.py:30-47).
"""
import numpy as np

__all__ = ['RandInterlv']


class _Interleaver:

    def interlv(self, in_array):
        """Gather: ``out[i] = in[p[i]]``."""
        return np.asarray(in_array)[self.p_array]

    def deinterlv(self, in_array):
        """Scatter: ``out[p[i]] = in[i]``."""
        in_array = np.asarray(in_array)
        out_array = np.zeros(len(in_array), in_array.dtype)
        out_array[self.p_array] = in_array
        return out_array


class RandInterlv(_Interleaver):
    """Random interleaver: ``RandomState(seed).permutation(arange(length))`` (interleavers.py:50-77),
    i.e. the legacy MT19937 stream, so the permutation matches the reference's for equal seeds."""

    def __init__(self, length, seed):
        self.p_array = np.random.RandomState(seed).permutation(np.arange(length))
