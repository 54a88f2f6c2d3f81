"""Monte-Carlo link simulation around the GPU decoders (host orchestration).

Mirrors /root/reference/commpy/links.py: ``LinkModel`` (links.py:67-343) with
``link_performance_full_metrics`` (:155-267) and ``link_performance`` (:269-343), and the module-level
``link_performance`` (:29-64).  The callback protocol is the reference's
(``modulate(bits)``, ``channel.propagate``, ``receive(y, H, constellation, noise_var)``,
``decoder(msg)``); the estimators (BER denominators incl. quirk B11, stop rules) are reproduced.

MI355X-first difference: the transmissions of one SNR point are *batched*.  When the callbacks are
marked batch-capable (attribute ``batched = True``; the ones built by ``commpy_amd.wifi80211`` are) a block
of ``tx_batch`` transmissions is generated, modulated, propagated, demodulated and decoded as 2-D
arrays -- one GPU launch per stage instead of one Python call per transmission -- and the reference's
sequential stop rule is applied to the per-transmission error counts afterwards.  As soon as one callback is
not marked, transmissions run one at a time exactly like the reference (same random stream, same stop rule,
six-argument decoders recognised by arity).  MIMO channels are out of scope.
"""
from fractions import Fraction
from inspect import getfullargspec

import numpy as np

__all__ = ['link_performance', 'LinkModel']


def link_performance(link_model, SNRs, send_max, err_min, send_chunk=None, code_rate=1):
    """Module-level convenience wrapper (links.py:29-64): a missing / zero chunk size means ``err_min`` bits per chunk."""
    return link_model.link_performance(SNRs, send_max, err_min, send_chunk or err_min, code_rate)


def _is_batched(fn):
    return bool(getattr(fn, 'batched', False))


class LinkModel:
    """Same constructor and attributes as links.py:67-153."""

    def __init__(self, modulate, channel, receive, num_bits_symbol, constellation, Es=1, decoder=None,
                 rate=Fraction(1, 1)):
        # the attribute set of links.py:137-153; a float rate becomes the nearest fraction with denominator <= 100
        self.rate = Fraction(rate).limit_denominator(100) if type(rate) is float else rate
        self.decoder = (lambda msg: msg) if decoder is None else decoder
        self.modulate, self.channel, self.receive = modulate, channel, receive
        self.num_bits_symbol, self.constellation, self.Es = num_bits_symbol, constellation, Es
        self.full_simulation_results = None
        self.tx_batch = 64            # transmissions generated / decoded per GPU batch

    # -- one block of transmissions ------------------------------------------------------------
    def _all_batched(self):
        return _is_batched(self.modulate) and _is_batched(self.receive) and _is_batched(self.decoder)

    def _block_size(self, remaining):
        """Transmissions per block: ``tx_batch`` on the batched (GPU) path; ONE when any callback is the reference's
        per-transmission kind, so that the sequential stop rule never runs (and discards) extra transmissions and the
        NumPy random stream is consumed exactly like the reference's."""
        return max(1, min(self.tx_batch if self._all_batched() else 1, remaining))

    def _run_block(self, n_tx, n_bits):
        """Returns (msgs [n_tx, n_bits], decoded [n_tx, >= n_bits]) for n_tx independent transmissions."""
        if self._all_batched():
            msgs = np.random.choice((0, 1), (n_tx, n_bits))
            symbs = self.modulate(msgs)
            out = self.channel.propagate(symbs)
            noise_var = self.channel.noise_std ** 2
            received = self.receive(out, self.channel.channel_gains, self.constellation, noise_var)
            return msgs, np.asarray(self.decoder(received))
        # per transmission, the reference's protocol (links.py:229-250), including the six-argument decoders
        # (``idd_decoder``-style, links.py:216, 246-248) that are recognised by their arity
        full_args_decoder = len(getfullargspec(self.decoder).args) > 1
        msgs, decs = [], []
        for _ in range(n_tx):
            msg = np.random.choice((0, 1), n_bits)
            out = self.channel.propagate(self.modulate(msg))
            noise_var = self.channel.noise_std ** 2
            received = self.receive(out, self.channel.channel_gains, self.constellation, noise_var)
            if full_args_decoder:
                dec = self.decoder(out, self.channel.channel_gains, self.constellation, noise_var, received,
                                   self.channel.nb_tx * self.num_bits_symbol)
            else:
                dec = self.decoder(received)
            msgs.append(msg)
            decs.append(np.asarray(dec).reshape(-1))
        return np.stack(msgs), np.stack(decs)

    def _prepare(self, send_chunk, err_min, code_rate):
        """Chunk size rounded down to a whole number of channel uses (at least one), code rate as a fraction (links.py:203-214)."""
        self.rate = code_rate = Fraction(code_rate).limit_denominator(100) if type(code_rate) is float else code_rate
        chunk = err_min if send_chunk is None else send_chunk
        divider = (Fraction(1, self.num_bits_symbol * self.channel.nb_tx) / code_rate).denominator
        return max(divider, chunk // divider * divider), code_rate

    def link_performance_full_metrics(self, SNRs, tx_max, err_min, send_chunk=None, code_rate=Fraction(1, 1),
                                      number_chunks_per_send=1, stop_on_surpass_error=True):
        """BER / bit errors / chunk errors / chunk counts per SNR and transmission (links.py:155-267).

        Stop rules as in the reference: a transmission is only counted while the accumulated bit
        errors of the SNR point do not exceed ``err_min`` (``stop_on_surpass_error``), and the sweep
        stops after the first SNR whose total errors stay below ``err_min``.  The BER denominator is
        ``total_tx_send * send_chunk`` -- it ignores ``number_chunks_per_send`` (quirk B11).
        """
        n_snr = len(SNRs)
        BERs = np.zeros_like(SNRs, dtype=float)
        BEs, CEs, NCs = (np.zeros((n_snr, tx_max), dtype=int) for _ in range(3))      # bit errors, chunk errors, chunk counts
        send_chunk, code_rate = self._prepare(send_chunk, err_min, code_rate)
        n_bits = send_chunk * number_chunks_per_send
        for id_SNR in range(len(SNRs)):
            self.channel.set_SNR_dB(SNRs[id_SNR], float(code_rate), self.Es)
            bit_err = np.zeros(tx_max, dtype=int)
            chunk_count = np.zeros(tx_max, dtype=int)
            total_tx_send = 0
            id_tx = 0
            stopped = False
            while id_tx < tx_max and not stopped:
                if stop_on_surpass_error and bit_err.sum() > err_min:   # tested BEFORE a transmission is made (:226)
                    break
                n_blk = self._block_size(tx_max - id_tx)
                msgs, dec = self._run_block(n_blk, n_bits)
                errs = (msgs != dec[:, :n_bits].astype(int)).reshape(n_blk, number_chunks_per_send, send_chunk).sum(2)
                for j in range(n_blk):                      # the reference's per-transmission bookkeeping
                    if stop_on_surpass_error and bit_err.sum() > err_min:
                        stopped = True
                        break
                    bit_err[id_tx] = errs[j].sum()
                    chunk_count[id_tx] = number_chunks_per_send
                    total_tx_send += 1
                    id_tx += 1
            n_err = int(bit_err.sum())
            BEs[id_SNR], NCs[id_SNR] = bit_err, chunk_count
            CEs[id_SNR] = bit_err > 0                        # a transmission is one chunk here: in error or not (:257)
            BERs[id_SNR] = n_err / (total_tx_send * send_chunk)              # quirk B11: ignores number_chunks_per_send
            if n_err < err_min:                              # this point never collected err_min errors: the sweep ends (:262)
                break
        self.full_simulation_results = results = (BERs, BEs, CEs, NCs)
        return results

    def link_performance(self, SNRs, send_max, err_min, send_chunk=None, code_rate=1):
        """BER per SNR: send chunks until ``send_max`` bits or ``err_min`` errors (links.py:269-343)."""
        send_chunk, code_rate = self._prepare(send_chunk, err_min, code_rate)
        curve = np.zeros(np.shape(SNRs), dtype=float)
        for point, snr_db in enumerate(SNRs):
            self.channel.set_SNR_dB(snr_db, float(code_rate), self.Es)
            sent = wrong = 0

            def running():                                   # the reference's loop condition (:323)
                return sent < send_max and wrong < err_min
            while running():
                n_blk = self._block_size(-(-(send_max - sent) // send_chunk))
                msgs, dec = self._run_block(n_blk, send_chunk)
                per_tx = (msgs != dec[:, :send_chunk].astype(int)).sum(1)
                for e in per_tx:                             # sequential stop rule: surplus transmissions of a block are dropped
                    if running():
                        sent, wrong = sent + send_chunk, wrong + int(e)
            curve[point] = wrong / sent
            if wrong < err_min:                              # send_max reached before err_min errors: higher SNRs stay 0 (:341)
                break
        return curve
