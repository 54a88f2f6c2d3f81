"""Bit helpers that define the bit order and metrics of the decoding hot path (host side).

Mirrors the public names and behaviour of the reference's ``commpy/utilities.py``
(/root/reference/commpy/utilities.py:30-154): MSB-first bit (un)packing, including the
wrap-around quirk of ``dec2bitarray`` for values that need more than ``bit_width`` bits
(utilities.py:81-85), Hamming and squared Euclidean distance.
These run on the host only; the HIP kernels consume the tables built from them.
"""
import numpy as np

__all__ = ['dec2bitarray', 'decimal2bitarray', 'bitarray2dec', 'hamming_dist', 'euclid_dist',
           'upsample', 'signal_power']


def decimal2bitarray(number, bit_width):
    """One non-negative integer -> ``int8[bit_width]``, MSB first (utilities.py:59-86).

    Bit ``p`` (LSB = 0) of ``number`` is written to ``result[bit_width - p - 1]``.  When
    ``p >= bit_width`` that index is negative and NumPy wraps it around, so bit ``p`` lands on
    the slot of bit ``p - bit_width`` (quirk B1: ``dec2bitarray(133, 7) == [0,0,0,0,1,0,1]``).
    Bits with ``p >= 2*bit_width`` raise ``IndexError`` exactly as in the reference.
    """
    number = int(number)
    result = np.zeros(bit_width, np.int8)
    p = 0
    while (1 << p) <= number:
        if (number >> p) & 1:
            result[bit_width - p - 1] = 1  # negative index wraps (or raises IndexError) like the reference
        p += 1
    return result


def dec2bitarray(in_number, bit_width):
    """Integer or array-like of integers -> concatenated MSB-first bit arrays (utilities.py:30-55)."""
    if isinstance(in_number, (np.integer, int)):
        return decimal2bitarray(in_number, bit_width)
    result = np.zeros(bit_width * len(in_number), np.int8)
    for pox, number in enumerate(in_number):
        result[pox * bit_width:(pox + 1) * bit_width] = decimal2bitarray(number, bit_width)
    return result


def bitarray2dec(in_bitarray):
    """MSB-first bit array -> Python/NumPy integer (utilities.py:89-109)."""
    number = 0
    for bit in in_bitarray:
        number = number * 2 + int(bit)
    return number


def hamming_dist(in_bitarray_1, in_bitarray_2):
    """Hamming distance = sum of the element-wise XOR (utilities.py:112-132)."""
    return np.bitwise_xor(in_bitarray_1, in_bitarray_2).sum()


def euclid_dist(in_array1, in_array2):
    """Squared Euclidean distance (utilities.py:135-154)."""
    d = np.asarray(in_array1) - np.asarray(in_array2)
    return (d * d).sum()


def upsample(x, n):
    """Insert ``n-1`` zeros between samples; complex output (utilities.py:157-182)."""
    y = np.zeros(len(x) * n, dtype=complex)
    y[0::n] = x
    return y


def signal_power(signal):
    """Mean of |s|^2 (utilities.py:185-205).

    Evaluated element by element on scalars like the reference's ``np.vectorize`` lambda: the scalar
    ``abs(s) ** 2`` and the array ``np.abs(x) ** 2`` differ in the last ulp (e.g. QAM-16: Es = 10.0
    vs 10.000000000000002), and Es feeds the noise calibration of the link simulations.
    """
    return np.mean(np.array([abs(s) ** 2 for s in np.asarray(signal).reshape(-1)]))
