"""CPU oracle of the CommPy decoding hot path -- TEST INFRASTRUCTURE ONLY.

ctypes wrapper around oracle/libcpx_oracle.so (C restatement in cpx_oracle.c, each function citing
the reference file:line it follows).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package, and only as the checker / CPU baseline.  The product
(commpy_amd/) never imports it.

Parity status: PINNED against the live reference through tests/golden/*.npz
(tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_int8, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcpx_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "cpx_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcpx_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB_PATH)
        lib.orc_np_sum.restype = c_double
        lib.orc_np_sum.argtypes = [c_void_p, c_int64, c_int64]
        lib.orc_dec2bitarray.argtypes = [c_int64, c_int, c_void_p]
        lib.orc_viterbi_decode.argtypes = [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                           c_int, c_int, c_void_p, POINTER(c_int64)]
        lib.orc_viterbi_decode_batch.argtypes = [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                 c_void_p, c_int, c_int, c_void_p, c_int64]
        lib.orc_map_decode.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_double,
                                       c_void_p, c_int, c_void_p, c_void_p]
        lib.orc_turbo_decode.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p,
                                         c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p]
        lib.orc_ldpc_bp_decode.argtypes = [c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, c_int,
                                           c_int, c_void_p, c_void_p, c_void_p]
        lib.orc_demod_soft.argtypes = [c_void_p, c_int64, c_void_p, c_int, c_int, c_double, c_void_p]
        lib.orc_demod_hard.argtypes = [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p]
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _tables(trellis):
    return (np.ascontiguousarray(trellis.next_state_table, dtype=np.int32),
            np.ascontiguousarray(trellis.output_table, dtype=np.int32))


_VIT = {'hard': 0, 'soft': 1, 'unquantized': 2}


def dec2bitarray(number, width):
    out = np.zeros(width, np.int8)
    if load().orc_dec2bitarray(int(number), int(width), _p(out)) != 0:
        raise IndexError("index out of bounds")
    return out


def viterbi_decode(coded_bits, trellis, tb_depth=None, decoding_type='hard'):
    """Oracle of convcode.py:661 for one codeword (1-D) or a batch (2-D, looped)."""
    lib = load()
    x = _f64(coded_bits)
    nxt, out = _tables(trellis)
    if x.ndim == 2:
        B, length = x.shape
        L = int(length * (trellis.k / trellis.n))
        dec = np.zeros((B, max(L, 1)), dtype=np.int64)
        rc = lib.orc_viterbi_decode_batch(_p(x), B, length, int(trellis.k), int(trellis.n), int(trellis.total_memory),
                                          int(trellis.number_states), int(trellis.number_inputs), _p(nxt), _p(out),
                                          0 if tb_depth is None else int(tb_depth), _VIT[decoding_type], _p(dec),
                                          dec.shape[1])
        if rc != 0:
            raise ValueError("oracle viterbi_decode failed (%d)" % rc)
        return dec[:, :L]
    L = int(len(x) * (trellis.k / trellis.n))
    dec = np.zeros(max(L, 1), dtype=np.int64)
    Lo = c_int64(0)
    rc = lib.orc_viterbi_decode(_p(x), len(x), int(trellis.k), int(trellis.n), int(trellis.total_memory),
                                int(trellis.number_states), int(trellis.number_inputs), _p(nxt), _p(out),
                                0 if tb_depth is None else int(tb_depth), _VIT[decoding_type], _p(dec),
                                ctypes.byref(Lo))
    if rc != 0:
        raise ValueError("oracle viterbi_decode failed (%d)" % rc)
    return dec[:L]


def viterbi_decode_mt(coded_bits, trellis, tb_depth=None, decoding_type='hard', threads=None):
    """``viterbi_decode`` of a 2-D batch spread over host threads (every slice is one C call, GIL released)."""
    import concurrent.futures as cf
    x = _f64(coded_bits)
    threads = threads or max(1, len(os.sched_getaffinity(0)))
    threads = max(1, min(threads, len(x)))
    parts = np.array_split(np.arange(len(x)), threads)
    with cf.ThreadPoolExecutor(threads) as ex:
        outs = list(ex.map(lambda idx: viterbi_decode(x[idx[0]:idx[-1] + 1], trellis, tb_depth, decoding_type), parts))
    return np.concatenate(outs, axis=0)


def map_decode(sys_symbols, non_sys_symbols, trellis, noise_variance, L_int, mode='decode'):
    """Oracle of turbo.py:163; returns [L_ext, decoded_bits]."""
    lib = load()
    s, p, li = _f64(sys_symbols), _f64(non_sys_symbols), _f64(L_int)
    N = len(s)
    nxt, out = _tables(trellis)
    L = np.zeros(N)
    bits = np.zeros(N, dtype=np.int64)
    rc = lib.orc_map_decode(_p(s), _p(p), N, int(trellis.n), int(trellis.number_states), int(trellis.number_inputs),
                            _p(nxt), _p(out), float(noise_variance), _p(li), 1 if mode == 'decode' else 0, _p(L),
                            _p(bits))
    if rc != 0:
        raise ValueError("oracle map_decode failed (%d)" % rc)
    return [L, bits]


def turbo_decode(sys_symbols, non_sys_symbols_1, non_sys_symbols_2, trellis, noise_variance, number_iterations,
                 interleaver, L_int=None):
    """Oracle of turbo.py:254."""
    lib = load()
    s, p1, p2 = _f64(sys_symbols), _f64(non_sys_symbols_1), _f64(non_sys_symbols_2)
    N = len(s)
    nxt, out = _tables(trellis)
    perm = np.ascontiguousarray(interleaver.p_array, dtype=np.int64)
    li = None if L_int is None else _f64(L_int)
    dec = np.zeros(N, dtype=np.int64)
    rc = lib.orc_turbo_decode(_p(s), _p(p1), _p(p2), N, int(trellis.n), int(trellis.number_states),
                              int(trellis.number_inputs), _p(nxt), _p(out), float(noise_variance),
                              int(number_iterations), _p(perm), None if li is None else _p(li), _p(dec))
    if rc != 0:
        raise ValueError("oracle turbo_decode failed (%d)" % rc)
    return dec


def ldpc_edges(ldpc_code_params):
    """Edge list sorted by (check, variable) from the reference's adjacency arrays (ldpc.py:39-41)."""
    n_c = int(ldpc_code_params['n_cnodes'])
    if 'cnode_adj_list' not in ldpc_code_params:                  # a dict that only carries the matrix (ldpc.py:189-195 reads just that)
        H = np.asarray(ldpc_code_params['parity_check_matrix'].todense()
                       if hasattr(ldpc_code_params['parity_check_matrix'], 'todense') else ldpc_code_params['parity_check_matrix'])
        ec, ev = np.nonzero(H)                                    # row-major: sorted by (check, variable)
        return ec.astype(np.int32), ev.astype(np.int32)
    mcd = int(ldpc_code_params['max_cnode_deg'])
    adj = np.asarray(ldpc_code_params['cnode_adj_list']).reshape(n_c, mcd)
    deg = np.asarray(ldpc_code_params['cnode_deg_list'])
    ec, ev = [], []
    for c in range(n_c):
        vs = np.unique(adj[c, :deg[c]])          # lil assignment = set semantics, columns sorted
        ec.append(np.full(len(vs), c, np.int32))
        ev.append(vs.astype(np.int32))
    return np.concatenate(ec), np.concatenate(ev)


def ldpc_bp_decode(llr_vec, ldpc_code_params, decoder_algorithm, n_iters, return_iters=False):
    """Oracle of ldpc.py:144 (clips llr_vec in place like the reference)."""
    lib = load()
    if decoder_algorithm not in ('SPA', 'MSA'):
        raise NameError('Please input a valid decoder_algorithm string (meanning "SPA" or "MSA").')
    n_v = int(ldpc_code_params['n_vnodes'])
    n_c = int(ldpc_code_params['n_cnodes'])
    ec, ev = ldpc_edges(ldpc_code_params)
    llr = _f64(llr_vec)
    nblk = llr.size // n_v
    dec = np.zeros(nblk * n_v, np.int8)
    out = np.zeros(nblk * n_v)
    its = np.zeros(nblk, np.int32)
    rc = lib.orc_ldpc_bp_decode(_p(llr), nblk, n_v, n_c, len(ec), _p(ec), _p(ev),
                                0 if decoder_algorithm == 'SPA' else 1, int(n_iters), _p(dec), _p(out), _p(its))
    if rc != 0:
        raise ValueError("oracle ldpc_bp_decode failed (%d)" % rc)
    if isinstance(llr_vec, np.ndarray) and llr_vec.dtype == np.float64:
        llr_vec[...] = llr.reshape(llr_vec.shape)   # in-place clip (ldpc.py:186)
    dec = dec.reshape(-1, nblk, order='F').squeeze().astype(np.int8)     # ldpc.py:251-253
    out = out.reshape(-1, nblk, order='F').squeeze()
    return (dec, out, its) if return_iters else (dec, out)


def demodulate(constellation, input_symbols, demod_type, noise_var=0):
    """Oracle of Modem.demodulate, modulation.py:100."""
    lib = load()
    c = np.ascontiguousarray(constellation, dtype=np.complex128)
    y = np.ascontiguousarray(np.atleast_1d(input_symbols), dtype=np.complex128)
    M = len(c)
    nb = int(np.log2(M))
    if demod_type == 'soft':
        out = np.zeros(len(y) * nb)
        lib.orc_demod_soft(_p(y), len(y), _p(c), M, nb, float(noise_var), _p(out))
        return out
    if demod_type == 'hard':
        out = np.zeros(len(y) * nb, np.int8)
        lib.orc_demod_hard(_p(y), len(y), _p(c), M, nb, _p(out))
        return out
    raise ValueError('demod_type must be "hard" or "soft"')
