"""GPU-resident coded link: the whole Monte-Carlo chain of BASELINE config 5 on the device.

``DeviceWifiLink`` runs, per batch of transmissions and without leaving HBM,

    random bits -> conv_encode('cont') -> puncturing -> modulate -> AWGN -> soft demodulation
                -> depuncturing -> soft Viterbi -> per-chunk error count

with the semantics of the reference's ``Wifi80211.link_performance`` chain
(/root/reference/commpy/wifi80211.py:132-216, links.py:155-267, channels.py:37-74), including quirk B7
(complex noise of per-component std ``noise_std/2`` while the demodulator is told ``noise_std**2``) and
quirk B1 (decimal generators) unless ``generator_matrix`` is given.  Only the error counters come back
to the host.  The random streams are Philox-based, so results are statistically -- not bit-wise --
comparable with the reference; the deterministic stages are bit-exact (tests/test_devicelink_gpu.py).

``DeviceBuf`` / the ``*_dev`` helpers are thin wrappers over the C-ABI for callers that keep data in HBM.
"""
import ctypes
import math

import numpy as np

from commpy_amd import _lib
from commpy_amd.wifi80211 import Wifi80211

__all__ = ['DeviceBuf', 'DeviceWifiLink', 'conv_encode_gpu', 'modulate_gpu']


class DeviceBuf:
    """A device allocation owned through the C-ABI (cpx_malloc / cpx_free)."""

    def __init__(self, nbytes):
        self.lib = _lib.load()
        self.nbytes = int(nbytes)
        self.ptr = ctypes.c_void_p()
        _lib.check(self.lib.cpx_malloc(ctypes.byref(self.ptr), max(self.nbytes, 8)))

    @classmethod
    def from_array(cls, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        if arr.nbytes:
            _lib.check(buf.lib.cpx_memcpy_h2d(buf.ptr, _lib.ptr(arr), arr.nbytes))
        return buf

    def to_array(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            _lib.check(self.lib.cpx_memcpy_d2h(_lib.ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.lib.cpx_free(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _encoded_length(nmsg, trellis, termination):
    """number_outbits of conv_encode (convcode.py:505-520)."""
    k, n, m = trellis.k, trellis.n, trellis.total_memory
    rate = float(k) / n
    if termination == 'cont':
        return int(nmsg / rate)
    if trellis.code_type == 'rsc':
        return int((nmsg + k * m) / rate)
    return int((nmsg + m + m % k) / rate)


def conv_encode_gpu(message_bits, trellis, termination='term'):
    """``conv_encode`` for a batch ``[B, nbits]`` on the GPU (no puncturing); returns int64 ``[B, nout]``."""
    lib = _lib.load()
    msgs = np.ascontiguousarray(np.atleast_2d(message_bits), dtype=np.uint8)
    B, nmsg = msgs.shape
    nout = _encoded_length(nmsg, trellis, termination)
    d_msg, d_out = DeviceBuf.from_array(msgs), DeviceBuf(B * nout)
    _lib.check(lib.cpx_conv_encode_batch_dev(trellis._device_handle(), d_msg.ptr, B, nmsg, int(termination != 'cont'),
                                             int(trellis.code_type == 'rsc'), d_out.ptr, nout, None))
    _lib.check(lib.cpx_stream_sync(None))
    return d_out.to_array((B, nout), np.uint8).astype(np.int64)


def modulate_gpu(modem, input_bits):
    """``Modem.modulate`` on the GPU; returns complex128 symbols."""
    lib = _lib.load()
    bits = np.ascontiguousarray(input_bits, dtype=np.uint8).reshape(-1)
    nsym = bits.size // modem.num_bits_symbol
    d_bits, d_sym = DeviceBuf.from_array(bits), DeviceBuf(nsym * 16)
    _lib.check(lib.cpx_modulate_dev(modem._device_handle(), d_bits.ptr, nsym, d_sym.ptr, None))
    _lib.check(lib.cpx_stream_sync(None))
    return d_sym.to_array((nsym,), np.complex128)


class DeviceWifiLink:
    """BER of an 802.11 MCS over AWGN, simulated entirely on the GPU.

    Parameters mirror ``Wifi80211``: ``mcs`` 0..9, optional ``generator_matrix`` (default: the reference's
    decimal ``(133, 171)``, quirk B1).  ``send_chunk`` is the frame length in information bits (rounded
    like links.py:212-214), ``frame_aggregation`` the number of frames per transmission.
    """

    def __init__(self, mcs, send_chunk=600, frame_aggregation=1, generator_matrix=None, seed=1):
        self.lib = _lib.load()
        _lib.require_device()
        self.wifi = Wifi80211(mcs, generator_matrix=generator_matrix)
        self.trellis = self.wifi._get_trellis()
        self.modem = self.wifi.get_modem()
        self.coding = self.wifi._get_coding()
        self.rate = self.coding[0] / self.coding[1]
        from fractions import Fraction
        divider = (Fraction(1, self.modem.num_bits_symbol) * 1 / Fraction(self.rate).limit_denominator(100)).denominator
        self.send_chunk = max(divider, send_chunk // divider * divider)
        self.agg = int(frame_aggregation)
        self.nbits = self.send_chunk * self.agg                       # information bits per transmission
        self.seed = int(seed)
        self._calls = 0
        # index maps of puncturing / depuncturing (vector form of convcode.py:752-804)
        self.ncoded = 2 * self.nbits                                  # rate-1/2 mother code, 'cont'
        pvec = Wifi80211._get_puncture_matrix(*self.coding)
        if pvec is None:
            self.keep_idx = None
            self.ntx = self.ncoded
            self.nde = self.ncoded
            self.de_idx = None
        else:
            pmask = np.asarray(pvec) == 1
            keep = pmask[np.arange(self.ncoded) % len(pmask)]
            self.keep_idx = np.flatnonzero(keep).astype(np.int32)
            self.ntx = len(self.keep_idx)
            self.nde = math.ceil(self.ntx * self.coding[0] / self.coding[1] * 2)
            keep2 = pmask[np.arange(self.nde) % len(pmask)]
            de = -np.ones(self.nde, dtype=np.int32)
            de[keep2] = np.arange(keep2.sum(), dtype=np.int32)
            if keep2.sum() > self.ntx:
                raise IndexError('depuncturing: message too short for the puncturing pattern')
            self.de_idx = de
        nb = self.modem.num_bits_symbol
        if self.ntx % nb:
            raise ValueError('send_chunk does not give an integer number of symbols')
        self.nsym = self.ntx // nb
        self._bufs = {}

    # -- buffers --------------------------------------------------------------------------------------------
    def _alloc(self, T):
        if self._bufs.get('T') == T:
            return self._bufs
        for b in self._bufs.values():
            if isinstance(b, DeviceBuf):
                b.free()
        nb = self.modem.num_bits_symbol
        bufs = {'T': T,
                'msg': DeviceBuf(T * self.nbits), 'coded': DeviceBuf(T * self.ncoded),
                'sym': DeviceBuf(T * self.nsym * 16), 'rx': DeviceBuf(T * self.nsym * 16),
                'llr': DeviceBuf(T * self.nsym * nb * 8), 'dec': DeviceBuf(T * self.nbits),
                'errs': DeviceBuf(T * self.agg * 4)}
        if self.keep_idx is not None:
            bufs['tx'] = DeviceBuf(T * self.ntx)
            bufs['llr_de'] = DeviceBuf(T * self.nde * 8)
            bufs['keep_idx'] = DeviceBuf.from_array(self.keep_idx)
            bufs['de_idx'] = DeviceBuf.from_array(self.de_idx)
        self._bufs = bufs
        return bufs

    # -- one batch of T transmissions at one SNR ----------------------------------------------------------------
    def run_batch(self, snr_db, T):
        """Simulate ``T`` transmissions; returns int32 ``[T, frame_aggregation]`` bit errors per frame."""
        lib, bufs = self.lib, self._alloc(T)
        nb = self.modem.num_bits_symbol
        h_tr, h_md = self.trellis._device_handle(), self.modem._device_handle()
        # channels.py:74 (complex channel): noise_std = sqrt(2 * Es / (rate * snr)); per-component std = noise_std/2
        noise_std = math.sqrt(2.0 * self.modem.Es / (self.rate * 10 ** (snr_db / 10.0)))
        self._calls += 1
        ck = _lib.check
        ck(lib.cpx_random_bits_dev(bufs['msg'].ptr, T * self.nbits, self.seed, 2 * self._calls, None))
        ck(lib.cpx_conv_encode_batch_dev(h_tr, bufs['msg'].ptr, T, self.nbits, 0, 0, bufs['coded'].ptr, self.ncoded, None))
        tx = bufs['coded']
        if self.keep_idx is not None:
            ck(lib.cpx_gather_u8_dev(bufs['coded'].ptr, T, self.ncoded, bufs['keep_idx'].ptr, self.ntx, bufs['tx'].ptr, None))
            tx = bufs['tx']
        ck(lib.cpx_modulate_dev(h_md, tx.ptr, T * self.nsym, bufs['sym'].ptr, None))
        ck(lib.cpx_awgn_dev(bufs['sym'].ptr, T * self.nsym, noise_std * 0.5, noise_std * 0.5, self.seed, 2 * self._calls + 1,
                            bufs['rx'].ptr, None))
        ck(lib.cpx_demod_soft_dev(h_md, bufs['rx'].ptr, T * self.nsym, noise_std ** 2, bufs['llr'].ptr, None))
        llr, length = bufs['llr'], self.ntx
        if self.keep_idx is not None:
            ck(lib.cpx_gather_f64_dev(bufs['llr'].ptr, T, self.ntx, bufs['de_idx'].ptr, self.nde, bufs['llr_de'].ptr, None))
            llr, length = bufs['llr_de'], self.nde
        m = self.trellis.total_memory
        L = int(length * 0.5)
        n_steps = int((L + m) / 1) - 1
        ck(lib.cpx_viterbi_decode_batch_dev(h_tr, llr.ptr, T, length, L, n_steps, min(5 * m, L), 1, bufs['dec'].ptr, None))
        ck(lib.cpx_count_errors_dev(bufs['msg'].ptr, self.nbits, bufs['dec'].ptr, L, T, self.agg, self.send_chunk,
                                    bufs['errs'].ptr, None))
        ck(lib.cpx_stream_sync(None))
        return bufs['errs'].to_array((T, self.agg), np.int32)

    def ber_sweep(self, snrs_db, n_bits, tx_batch=4096):
        """BER per SNR over at least ``n_bits`` information bits each (no early stopping)."""
        out = []
        for snr in snrs_db:
            done, errs = 0, 0
            while done < n_bits:
                T = int(min(tx_batch, math.ceil((n_bits - done) / self.nbits)))
                e = self.run_batch(float(snr), T)
                errs += int(e.sum())
                done += T * self.nbits
            out.append(errs / done)
        return np.array(out)
