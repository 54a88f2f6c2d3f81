"""Convolutional codes: host-side code description + MI355X Viterbi decoder.

Public names and argument meaning follow the reference module
/root/reference/commpy/channelcoding/convcode.py:

* ``Trellis``          (convcode.py:23-255)  host: builds next_state_table / output_table;
* ``conv_encode``      (convcode.py:475-558) host: table-driven encoder (input generator);
* ``puncturing`` / ``depuncturing`` (convcode.py:752-804) host;
* ``viterbi_decode``   (convcode.py:661-749) DEVICE: the body is the HIP kernel
  ``viterbi_wave_kernel`` (commpy_amd/csrc/viterbi.hip) reached through the C-ABI
  ``cpx_viterbi_decode_batch`` (include/commpy_amd.h).  A 2-D input ``[B, len]`` decodes a
  batch of independent codewords (extension; the reference is 1-D only).

The kernels are table-driven: they never re-derive tables from polynomials (the legacy-int and the
matrix feedback constructions give different RSC output tables).
"""
import ctypes
import os
import math
from warnings import warn

import numpy as np

from commpy_amd import _lib
from commpy_amd.utilities import dec2bitarray, bitarray2dec

__all__ = ['Trellis', 'conv_encode', 'conv_encode_batch', 'viterbi_decode', 'puncturing', 'depuncturing']

_VIT_TYPES = {'hard': 0, 'soft': 1, 'unquantized': 2}


def _poly_taps(value, width, msb_format):
    """Taps of one polynomial as an int array ``taps[w]`` = coefficient of D^w (convcode.py:211-222).

    'MSB' format: the least significant bit of the number is the D^0 tap (the reference reverses
    ``dec2bitarray(value, width)``); 'LSB'/'Matlab': the most significant of the ``width`` bits is.
    ``dec2bitarray`` keeps its wrap-around quirk for numbers wider than ``width`` bits.
    """
    bits = dec2bitarray(int(value), width)
    return bits[::-1].copy() if msb_format else bits.copy()


class Trellis:
    """Trellis of a k/n convolutional code (same constructor as convcode.py:117).

    Attributes: ``k, n, total_memory, number_states, number_inputs, next_state_table,
    output_table, code_type`` -- identical meaning and values to the reference
    (golden tables: commpy/channelcoding/tests/test_convcode.py:23-111).

    State numbering: the state integer is the concatenation (MSB first) of the shift registers,
    most recent bit first; ``output_table[s, i]`` packs the n output bits MSB = output 0.
    """

    def __init__(self, memory, g_matrix, feedback=None, code_type='default', polynomial_format='MSB'):
        memory = np.asarray(memory)
        k, n = g_matrix.shape
        m = int(memory.sum())
        self.k, self.n, self.total_memory, self.code_type = k, n, m, code_type
        self.number_states, self.number_inputs = 1 << m, 1 << k
        self.next_state_table, self.output_table = (np.zeros((1 << m, 1 << k), dtype=int) for _ in range(2))

        if isinstance(feedback, int):
            warn('Trellis  will only accept feedback as a matrix in the future. '
                 'Using the backwards compatibility version that may contain bugs for k > 1 or with LSB format.',
                 DeprecationWarning)
            self._build_legacy(memory, g_matrix, feedback)
        else:
            self._build_matrix(memory, g_matrix, feedback, polynomial_format)

    # -- matrix construction, convcode.py:195-255 -------------------------------------------
    def _build_matrix(self, memory, g_matrix, feedback, polynomial_format):
        if polynomial_format == 'MSB':
            msb = True
        elif polynomial_format in ('LSB', 'Matlab'):
            msb = False
        else:
            raise ValueError('polynomial_format must be "LSB", "MSB" or "Matlab"')
        k, n = self.k, self.n
        width = int(memory.max()) + 1
        if feedback is None:
            feedback = np.identity(k, int)
            if not msb:
                feedback = feedback * 2 ** int(memory.max())
        feedback = np.asarray(feedback)
        # fb[w, i, j]: tap on D^w of feedback polynomial (i, j);  gen[w, i, j]: same for g_matrix
        fb = np.zeros((width, k, k), np.int64)
        gen = np.zeros((width, k, n), np.int64)
        for i in range(k):
            for j in range(k):
                fb[:, i, j] = _poly_taps(feedback[i, j], width, msb)
            for j in range(n):
                gen[:, i, j] = _poly_taps(g_matrix[i, j], width, msb)
        offsets = np.concatenate(([0], np.cumsum(memory)))[:-1]
        for state in range(self.number_states):
            state_bits = dec2bitarray(state, self.total_memory).astype(np.int64)
            for inp in range(self.number_inputs):
                # regs[w, l]: row 0 = current input of register l, rows 1..mem_l = its delay line
                regs = np.zeros((width, k), np.int64)
                regs[0, :] = dec2bitarray(inp, k)
                for l, mem in enumerate(memory):
                    regs[1:mem + 1, l] = state_bits[offsets[l]:offsets[l] + mem]
                # outputs use the RAW input (before feedback), convcode.py:243-244
                out_bits = np.einsum('wl,wlj->j', regs, gen) % 2
                self.output_table[state, inp] = bitarray2dec(out_bits)
                # feedback-modified register inputs, convcode.py:247-248
                new_in = np.einsum('wl,wjl->j', regs, fb) % 2
                regs[0, :] = new_in
                nxt = state_bits.copy()
                for l, mem in enumerate(memory):
                    nxt[offsets[l]:offsets[l] + mem] = regs[:mem, l]
                self.next_state_table[state, inp] = bitarray2dec(nxt)

    # -- legacy construction (integer feedback), convcode.py:130-193 ---------------------------
    def _build_legacy(self, memory, g_matrix, feedback):
        k, n = self.k, self.n
        if self.code_type == 'rsc':
            for i in range(k):
                g_matrix[i][i] = feedback  # the reference mutates the caller's g_matrix (:135-137)
        if k != 1:
            # The reference's k > 1 legacy branch multiplies arrays of mismatching lengths
            # (convcode.py:166-168) and fails with a broadcasting ValueError.
            raise ValueError('legacy integer feedback is only usable for k = 1; pass feedback as a matrix')
        m = int(memory[0])
        fb_bits = dec2bitarray(feedback, m + 1).astype(np.int64)
        for state in range(self.number_states):
            for inp in range(self.number_inputs):
                outbits = np.zeros(n, np.int64)
                in_bit = int(dec2bitarray(inp, k)[0])
                sr = None
                for r in range(n):
                    sr = dec2bitarray(state, self.total_memory).astype(np.int64)
                    gen = dec2bitarray(g_matrix[0][r], m + 1).astype(np.int64)
                    acc = int((sr[:m] * gen[1:m + 1]).sum()) % 2          # delay-line taps (:154-156)
                    fa = int((fb_bits[1:] * sr[0:m]).sum())               # feedback sum (:160)
                    sr[1:m] = sr[0:m - 1].copy()                           # shift (:161-162)
                    sr[0] = (in_bit + fa) % 2                              # (:163-164)
                    outbits[r] = (acc + ((in_bit * int(gen[0]) + fa) % 2)) % 2   # (:175-177)
                self.output_table[state][inp] = bitarray2dec(outbits)
                self.next_state_table[state][inp] = bitarray2dec(sr)

    # -- any trellis-like object ------------------------------------------------------------------
    @classmethod
    def from_tables(cls, k, n, total_memory, next_state_table, output_table, code_type='default'):
        """A Trellis from ready-made tables (extension): what the decoders read is ``k, n, total_memory, number_states,
        number_inputs, next_state_table, output_table`` (convcode.py:590-749, turbo.py:78-158), so any code that can be
        written as such tables decodes -- also one the reference's constructor cannot build."""
        self = cls.__new__(cls)
        self.k, self.n, self.total_memory, self.code_type = int(k), int(n), int(total_memory), code_type
        self.next_state_table = np.array(next_state_table, dtype=int)
        self.output_table = np.array(output_table, dtype=int)
        self.number_states, self.number_inputs = self.next_state_table.shape
        if self.number_inputs != 2 ** self.k or self.output_table.shape != self.next_state_table.shape:
            raise ValueError('tables must be [number_states, 2 ** k]')
        return self

    # -- device handle ---------------------------------------------------------------------------
    def _device_handle(self):
        """Opaque cpx_trellis* carrying the tables to the current GPU (created on first use, one per device)."""
        hs = self.__dict__.get('_cpx_handles')
        if hs is None:
            def create():
                nxt = _lib.as_i32(self.next_state_table)
                out = _lib.as_i32(self.output_table)
                h = ctypes.c_void_p()
                _lib.check(_lib.load().cpx_trellis_create(int(self.k), int(self.n), int(self.number_states),
                                                          int(self.number_inputs),
                                                          nxt.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                          out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                          ctypes.byref(h)))
                if self.__dict__.get('_cpx_specialize') or os.environ.get('CPX_VITERBI_JIT', '0') not in ('', '0'):
                    from commpy_amd import jit
                    jit.specialize_trellis(h)                     # (no compiler / nothing to gain: the handle stays as it is)
                return h
            hs = self.__dict__['_cpx_handles'] = _lib.DeviceHandles(create, 'cpx_trellis_destroy')
        return hs.get()

    def specialize(self):
        """Compile (once, cached on disk) and attach the fused Viterbi kernels for THIS code's generators -- extension, no reference
        counterpart (commpy_amd/jit.py).  Rate-1/2 codes of full constraint length that are not among the built-in pairs otherwise
        run a table-driven kernel that is ~15 % slower on large batches; the decoded bits are identical.  Returns True when the
        current device's handle carries a code object afterwards; handles created later (other devices) follow suit."""
        from commpy_amd import jit
        self.__dict__['_cpx_specialize'] = True
        hs = self.__dict__.get('_cpx_handles')
        fresh = hs is None
        h = self._device_handle()                                  # (a handle created here has been specialised already)
        if not fresh:
            jit.specialize_trellis(h)
        return bool(jit.has_code_object(h))


def device_trellis(trellis):
    """cpx_trellis* for ``trellis``: a commpy_amd Trellis, or ANY object with the reference's trellis attributes (the
    reference's decoders are duck-typed, convcode.py:590-749) -- its tables are uploaded once and cached on the object."""
    own = getattr(trellis, '_device_handle', None)
    if own is not None:
        return own()
    cached = getattr(trellis, '_cpx_shadow', None)
    if cached is None or cached[0] is not trellis.next_state_table or cached[1] is not trellis.output_table:
        shadow = Trellis.from_tables(trellis.k, trellis.n, trellis.total_memory, trellis.next_state_table,
                                     trellis.output_table, getattr(trellis, 'code_type', 'default'))
        cached = (trellis.next_state_table, trellis.output_table, shadow)
        try:
            trellis._cpx_shadow = cached
        except AttributeError:
            pass
    return cached[2]._device_handle()


def conv_encode(message_bits, trellis, termination='term', puncture_matrix=None):
    """Table-driven convolutional encoder (host) -- convcode.py:475-558.

    One codeword is a batch of one: the table walk is ``conv_encode_batch``'s (a gather per trellis step), and what is
    left here is the reference's puncturing behaviour -- only row 0 of ``puncture_matrix`` is consulted, indexed by flat
    output position modulo the row length, and the survivors are packed to the front of a buffer that keeps the
    unpunctured length, i.e. the result ends in a zero tail (quirk B3, convcode.py:523-527, 552-556).
    """
    stream = conv_encode_batch(np.reshape(message_bits, (1, -1)), trellis, termination)[0]
    if puncture_matrix is None:
        return stream.astype(int)
    pattern = np.asarray(puncture_matrix)[0]
    survivors = stream[np.resize(pattern, stream.size) == 1]
    return np.concatenate([survivors, np.zeros(stream.size - survivors.size, stream.dtype)]).astype(int)


def conv_encode_batch(message_bits, trellis, termination='term'):
    """Vectorised host encoder for a batch ``[B, nbits]`` of messages (extension; no puncturing).

    Row ``b`` equals ``conv_encode(message_bits[b], trellis, termination)`` (convcode.py:475-558): the
    table walk is done for all codewords at once, one NumPy gather per trellis step.  Used to
    synthesise the large benchmark batches (the per-codeword Python loop of ``conv_encode`` needs
    minutes for 65536 codewords).
    """
    k, n, total_memory = trellis.k, trellis.n, trellis.total_memory
    msgs = np.asarray(message_bits).astype(np.int64)
    if msgs.ndim != 2:
        raise ValueError('message_bits must be [B, nbits]')
    B, nmsg = msgs.shape
    rsc = trellis.code_type == 'rsc' and termination != 'cont'
    rsc_term = rsc and termination == 'term'                  # the tail is clocked for 'term' only (:538), whereas ANY
    if termination == 'cont' or rsc:                          # termination but 'cont' reserves room for it (:509-512)
        inbits = msgs
    else:
        pad = total_memory + total_memory % k
        inbits = np.concatenate([msgs, np.zeros((B, pad), np.int64)], axis=1)
    nsteps = int(inbits.shape[1] / k)
    weights = 1 << np.arange(k - 1, -1, -1)
    nxt = np.asarray(trellis.next_state_table)
    outt = np.asarray(trellis.output_table)
    state = np.zeros(B, np.int64)
    symbols = []
    for i in range(nsteps):
        cur = inbits[:, i * k:(i + 1) * k].dot(weights)
        symbols.append(outt[state, cur])
        state = nxt[state, cur]
    if rsc_term:
        # term_bits = dec2bitarray(final_state, total_memory)[::-1], taken ONCE after the message (:539-540)
        tb = (state[:, None] >> np.arange(total_memory)) & 1
        for i in range(total_memory):
            grp = tb[:, i * k:(i + 1) * k]
            cur = grp.dot(1 << np.arange(grp.shape[1] - 1, -1, -1))
            symbols.append(outt[state, cur])
            state = nxt[state, cur]
    sym = np.stack(symbols, axis=1) if symbols else np.zeros((B, 0), np.int64)   # [B, steps]
    bits = ((sym[:, :, None] >> np.arange(n - 1, -1, -1)) & 1).reshape(B, -1)     # MSB-first n bits per step
    nout = int((nmsg + k * total_memory) / (float(k) / n)) if rsc else bits.shape[1]
    if nout > bits.shape[1]:                                         # e.g. termination='rsc' (turbo.py:47): zero tail
        bits = np.concatenate([bits, np.zeros((B, nout - bits.shape[1]), bits.dtype)], axis=1)
    return bits[:, :nout].astype(np.int64)


def puncture_keep_mask(n_positions, punct_vec):
    """Boolean mask of the positions ``puncturing`` keeps / ``depuncturing`` fills, for a sequence of ``n_positions``.

    The reference walks the positions with a running ``shift`` that grows after every position ``idx`` with
    ``idx % N == 0`` and looks up ``punct_vec[idx - shift * N]`` (convcode.py:766-772, :795-802).  At position ``idx``
    the shift is the number of earlier multiples of ``N``, ``ceil(idx / N)``, so the looked-up index lies in
    ``(-N, 0]`` and wraps like any negative Python index.  This is that rule as one index table."""
    pv = np.asarray(punct_vec)
    N = len(pv)
    idx = np.arange(int(n_positions))
    return pv[idx - -(-idx // N) * N] == 1


def puncturing(message, punct_vec):
    """Drop the positions whose puncture-vector entry is 0 -- convcode.py:752-774."""
    message = np.asarray(message)
    return message[puncture_keep_mask(len(message), punct_vec)]


def depuncturing(punctured, punct_vec, shouldbe):
    """Re-insert zeros at punctured positions -- convcode.py:777-804; float64 array of length ``shouldbe``.
    ``IndexError`` when ``punctured`` is too short for the pattern, like the reference's ``punctured[idx - shift2]``."""
    keep = puncture_keep_mask(shouldbe, punct_vec)
    punctured = np.asarray(punctured)
    nkeep = int(keep.sum())
    if nkeep > len(punctured):
        raise IndexError('index %d is out of bounds for axis 0 with size %d' % (len(punctured), len(punctured)))
    depunctured = np.zeros((int(shouldbe),))
    depunctured[keep] = punctured[:nkeep].astype(float)
    return depunctured


def _viterbi_sizes(length, trellis, tb_depth):
    """(L, n_steps, tb_depth) exactly as convcode.py:694-702, 721 computes them."""
    k, n = trellis.k, trellis.n
    rate = k / n
    L = int(length * rate)
    if tb_depth is None:
        tb_depth = min(5 * trellis.total_memory, L)
    n_steps = int((L + trellis.total_memory) / k) - 1
    return L, n_steps, int(tb_depth)


def viterbi_decode(coded_bits, trellis, tb_depth=None, decoding_type='hard'):
    """Viterbi decoding on MI355X; same signature/return as convcode.py:661.

    Parameters are those of the reference.  ``coded_bits`` may also be ``[B, len]`` (batch of
    independent codewords -> ``[B, L]``).  Returns int64 bits of length ``L = int(len*k/n)``,
    tail included (convcode.py:749).  'soft' inputs are LLRs log P1/P0, clipped to +-500.

    Differences kept on purpose: the caller's array is never modified (the reference overwrites
    the last received codeword with the padding value for 'hard'/'unquantized', convcode.py:724-732);
    an invalid ``decoding_type`` raises ``ValueError`` immediately rather than only when the padding
    steps are reached (convcode.py:733-734); a ``tb_depth`` beyond the number of trellis steps + 1 -- where the
    reference never runs a traceback and returns its uninitialised ``np.empty`` buffer (convcode.py:711, :644) --
    decodes with one full-length traceback from the best final state instead of returning garbage.  Abnormal
    inputs decode exactly like the reference: non-binary 'hard' values, +-inf, and NaN -- which the reference's
    clip lets through (convcode.py:719): from the first NaN step of a codeword on every metric is NaN, every
    decision is "first predecessor" and every traceback starts from state 0 (the fast kernels detect the NaN and
    such codewords are decoded a second time by a NaN-exact kernel).
    """
    if decoding_type not in _VIT_TYPES:
        raise ValueError('The available decoding types are "hard", "soft" and "unquantized')
    lib = _lib.load()
    arr = np.asarray(coded_bits)
    single = arr.ndim == 1
    x = _lib.as_f64(arr.reshape(1, -1) if single else arr)
    if x.ndim != 2:
        raise ValueError('coded_bits must be 1-D or 2-D [batch, len]')
    B, length = x.shape
    L, n_steps, tb = _viterbi_sizes(length, trellis, tb_depth)
    # int64 like the reference's `decoded_bits`; widened on the device (doing it with astype on the host costs more
    # than the decode for large batches)
    res = np.empty((B, L), dtype=np.int64)                   # every element is written by the library
    if B and L:
        if tb < 2:
            raise ValueError('tb_depth must be >= 2')
        _lib.check(lib.cpx_viterbi_decode_batch_i64(device_trellis(trellis), _lib.ptr(x), B, length, L, n_steps, tb,
                                                    _VIT_TYPES[decoding_type], _lib.ptr(res)))
    return res[0] if single else res
