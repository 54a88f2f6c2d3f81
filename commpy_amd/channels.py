"""Flat SISO channel models used to drive the decoders (host side, NumPy RNG).

Mirrors the part of /root/reference/commpy/channels.py the decoding path uses:
``SISOFlatChannel`` (channels.py:99-240, with ``_FlatChannel.set_SNR_dB`` :57-74 and
``generate_noises`` :37-55) and the ``bec`` / ``bsc`` / ``awgn`` helpers (:630-708).
MIMO channels are out of scope.  ``propagate`` also accepts a 2-D
``[batch, nsym]`` message so that a whole Monte-Carlo batch is generated at once.

Kept quirk B7: for a complex channel the generated noise is
``(randn + 1j*randn) * noise_std * 0.5`` (channels.py:53) while receivers are told
``noise_std**2`` (links.py:242-243).
"""
from numpy import abs, absolute, asarray, isrealobj, sqrt, where, zeros
from numpy.random import randn, random, standard_normal

__all__ = ['SISOFlatChannel', 'bec', 'bsc', 'awgn']


class SISOFlatChannel:
    """AWGN / Rice / Rayleigh flat SISO channel -- same constructor and attributes as channels.py:99."""

    def __init__(self, noise_std=None, fading_param=(1, 0)):
        self.noises = None
        self.channel_gains = None
        self.unnoisy_output = None
        self.noise_std = noise_std
        self.fading_param = fading_param

    nb_tx = nb_rx = property(lambda self: 1, doc='one antenna on each side (channels.py:223-231)')

    @property
    def isComplex(self):
        return self._isComplex

    @property
    def fading_param(self):
        return self._fading_param

    @fading_param.setter
    def fading_param(self, value):
        mean, variance = value[0], value[1]
        if variance + absolute(mean) ** 2 != 1:                      # energy conservation test of channels.py:205-206
            raise ValueError('fading_param = (mean, variance) must satisfy variance + |mean|^2 == 1: this channel would '
                             'add or remove energy')
        self._isComplex = isinstance(mean, complex)
        self._fading_param = value

    @property
    def k_factor(self):
        return absolute(self.fading_param[0]) ** 2 / absolute(self.fading_param[1])

    def set_SNR_dB(self, SNR_dB, code_rate=1., Es=1):
        """noise_std = sqrt((isComplex + 1) * nb_tx * Es / (code_rate * 10^(SNR/10)))  (channels.py:74)."""
        self.noise_std = sqrt((self.isComplex + 1) * self.nb_tx * Es / (code_rate * 10 ** (SNR_dB / 10)))

    def set_SNR_lin(self, SNR_lin, code_rate=1, Es=1):
        self.noise_std = sqrt((self.isComplex + 1) * self.nb_tx * Es / (code_rate * SNR_lin))

    def generate_noises(self, dims):
        """Noise samples for one propagation (channels.py:37-55).  Draw order and scaling are the reference's: real part
        first, and HALF of noise_std per component on the complex branch (:53) -- the receiver is told noise_std**2."""
        if self.noise_std is None:
            raise AssertionError('Noise standard deviation must be set before propagation.')
        noises = standard_normal(dims)
        if self.isComplex:
            noises = (noises + 1j * standard_normal(dims)) * self.noise_std * 0.5
        else:
            noises = noises * self.noise_std
        self.noises = noises

    def propagate(self, msg):
        """Fading + noise (channels.py:181-221); ``msg`` may be 1-D or ``[batch, nsym]``."""
        msg = asarray(msg)
        cplx = self.isComplex
        if not (cplx or isrealobj(msg)):
            raise TypeError('a complex message cannot be propagated in a real channel.')
        dims = msg.shape
        self.generate_noises(dims)                             # draw order of the reference: noise first, then the fading
        mean, variance = self.fading_param
        if cplx:
            scatter = (standard_normal(dims) + 1j * standard_normal(dims)) * sqrt(0.5 * variance)
        else:
            scatter = standard_normal(dims) * sqrt(variance)
        gains = self.channel_gains = mean + scatter
        clean = self.unnoisy_output = gains * msg
        return clean + self.noises


def bec(input_bits, p_e):
    """Binary erasure channel: erased bits become -1 (channels.py:630-649)."""
    output_bits = asarray(input_bits).copy()
    output_bits[random(len(output_bits)) <= p_e] = -1
    return output_bits


def bsc(input_bits, p_t):
    """Binary symmetric channel (channels.py:652-673)."""
    bits = asarray(input_bits)
    return where(random(len(bits)) <= p_t, 1 ^ bits, bits)          # one uniform draw per bit, flipped where it is <= p_t


def awgn(input_signal, snr_dB, rate=1.0):
    """White Gaussian noise at ``snr_dB`` relative to the signal's own mean energy (channels.py:676-708).

    The noise power per real dimension is ``mean(|x|^2) / (2 * rate * 10^(snr_dB / 10))``; a complex signal gets that on each axis,
    a real one twice that on its single axis.  The draws come from NumPy's global generator in the reference's order (all real
    parts, then all imaginary parts) and the mean energy is NumPy's pairwise ``sum`` (the reference module imports ``sum`` from
    numpy), so a seeded call returns the reference's samples bit for bit (tests/test_links_host.py)."""
    x = asarray(input_signal)
    n = len(x)
    energy = abs(x) * abs(x)
    per_axis = (energy.sum() / n) / (2 * rate * 10 ** (snr_dB / 10.0))
    if isinstance(x[0], complex):                                  # (the reference's own type test: element 0 decides)
        sigma = sqrt(per_axis)
        re, im = randn(n), randn(n)
        return x + (sigma * re + sigma * im * 1j)
    return x + sqrt(2 * per_axis) * randn(n)
