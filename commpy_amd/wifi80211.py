"""802.11 convolutionally-coded link (BASELINE config 5) on the GPU decoders.

Mirrors /root/reference/commpy/wifi80211.py: ``Wifi80211(mcs)`` with ``get_modem`` (:51-73),
``_get_puncture_matrix`` (:75-89), ``_get_coding`` (:91-104), ``_get_trellis`` (:106-108) and
``link_performance`` (:132-216).  The chain per transmission is the reference's:
conv_encode('cont') -> puncturing -> modulate -> channel -> soft demodulation (GPU) -> depuncturing
-> soft Viterbi (GPU) -> per-chunk error count; here a whole block of transmissions runs through
every stage as one 2-D array.

Quirk B1 is kept by default: the reference passes the generators in DECIMAL, ``(133, 171)``, and
``dec2bitarray`` silently wraps them to ``(5, 43)`` -- a catastrophic code.  Pass
``generator_matrix=np.array([[0o133, 0o171]])`` to simulate the code the standard means.
"""
import math

import numpy as np

import commpy_amd.channelcoding as cc
import commpy_amd.links as lk
import commpy_amd.modulation as mod

__all__ = ['Wifi80211']


def _batched(fn):
    fn.batched = True
    return fn


class Wifi80211:
    memory = np.array(6, ndmin=1)
    generator_matrix = np.array((133, 171), ndmin=2)     # decimal, exactly as wifi80211.py:49 (quirk B1)
    # transmissions generated / demodulated / decoded per GPU batch (None: LinkModel's default).  1 reproduces the reference's
    # sequence draw for draw -- one message, one noise vector at a time from NumPy's global stream -- and with it the reference's
    # per-transmission error counts under the same seed (tests/test_wifi_gpu.py pins that against tests/golden/wifi.npz)
    tx_batch = None

    def __init__(self, mcs, generator_matrix=None):
        self.mcs = mcs
        self.modem = None
        self.model = None
        if generator_matrix is not None:
            self.generator_matrix = np.array(generator_matrix, ndmin=2)

    def get_modem(self):
        """Modem of the MCS (wifi80211.py:51-73): PSK for MCS 0-2, square QAM above."""
        psk, qam = mod.PSKModem, mod.QAMModem
        family, order = ((psk, 2), (psk, 4), (psk, 4), (qam, 16), (qam, 16), (qam, 64), (qam, 64), (qam, 64), (qam, 256),
                         (qam, 256))[self.mcs]
        return family(order)

    @staticmethod
    def _get_puncture_matrix(numerator, denominator):
        """Puncturing vectors of 802.11-2016 (wifi80211.py:75-89)."""
        return {(2, 3): [1, 1, 1, 0], (3, 4): [1, 1, 1, 0, 0, 1],
                (5, 6): [1, 1, 1, 0, 0, 1, 1, 0, 0, 1]}.get((numerator, denominator))

    def _get_coding(self):
        coding = [(1, 2), (1, 2), (3, 4), (1, 2), (3, 4), (2, 3), (3, 4), (5, 6), (3, 4), (5, 6)]
        return coding[self.mcs]

    def _get_trellis(self):
        return cc.Trellis(self.memory, self.generator_matrix)

    def link_performance(self, channel, SNRs, tx_max, err_min, send_chunk=None, frame_aggregation=1, receiver=None,
                         stop_on_surpass_error=True):
        """Monte-Carlo BER of the link; same arguments and return as wifi80211.py:132
        (``BERs, BEs, CEs, NCs`` of ``LinkModel.link_performance_full_metrics``)."""
        trellis1 = self._get_trellis()
        coding = self._get_coding()
        modem = self.get_modem()
        self.modem = modem
        pvec = self._get_puncture_matrix(coding[0], coding[1])

        # Every closure accepts the reference's 1-D arrays (one transmission -> 1-D result, wifi80211.py:178-206) and
        # 2-D [T, n] blocks (-> 2-D); LinkModel only batches when ALL callbacks are marked, so a user-supplied
        # per-transmission `receiver` sees exactly what the reference would hand it.
        @_batched
        def modulate(bits):                                  # bits [n] or [T, n] -> symbols [nsym] or [T, nsym]
            bits = np.asarray(bits)
            res = cc.conv_encode_batch(np.atleast_2d(bits), trellis1, 'cont')
            if pvec is not None:
                res = res[:, cc.convcode.puncture_keep_mask(res.shape[1], pvec)]
            sym = modem.modulate(res.reshape(-1)).reshape(res.shape[0], -1)
            return sym[0] if bits.ndim == 1 else sym

        @_batched
        def _receiver(y, h, constellation, noise_var):       # soft LLRs on the GPU, [nsym*nb] or [T, nsym*nb]
            y = np.asarray(y)
            llr = modem.demodulate(y.reshape(-1), 'soft', noise_var)
            return llr if y.ndim == 1 else llr.reshape(y.shape[0], -1)

        if not receiver:
            receiver = _receiver

        @_batched
        def decoder_soft(msg):
            msg = np.asarray(msg)
            single = msg.ndim == 1
            msg = np.atleast_2d(msg)
            if pvec is not None:                             # depuncturing: zeros at the punctured positions
                shouldbe = math.ceil(msg.shape[1] * coding[0] / coding[1] * 2)
                keep = cc.convcode.puncture_keep_mask(shouldbe, pvec)
                if keep.sum() > msg.shape[1]:
                    raise IndexError('depuncturing: message too short for the puncturing pattern')
                full = np.zeros((msg.shape[0], shouldbe))
                full[:, keep] = msg[:, :keep.sum()]
                msg = full
            dec = cc.viterbi_decode(msg, trellis1, decoding_type='soft')
            return dec[0] if single else dec

        self.model = lk.LinkModel(modulate, channel, receiver, modem.num_bits_symbol, modem.constellation, modem.Es,
                                  decoder_soft, coding[0] / coding[1])
        if self.tx_batch is not None:
            self.model.tx_batch = int(self.tx_batch)
        return self.model.link_performance_full_metrics(SNRs, tx_max, err_min=err_min, send_chunk=send_chunk,
                                                        code_rate=coding[0] / coding[1],
                                                        number_chunks_per_send=frame_aggregation,
                                                        stop_on_surpass_error=stop_on_surpass_error)
