"""LDPC codes: host design-file handling + MI355X belief-propagation decoder.

Same public names, arguments and return conventions as /root/reference/commpy/channelcoding/ldpc.py:

* ``get_ldpc_code_params`` (ldpc.py:51-141), ``build_matrix`` (ldpc.py:13-48),
  ``write_ldpc_params`` (ldpc.py:257-299), ``triang_ldpc_systematic_encode`` (ldpc.py:302-354): host;
* ``ldpc_bp_decode`` (ldpc.py:144-254): DEVICE -> ``cpx_ldpc_bp_decode_batch`` (csrc/ldpc.hip).

The design-file text format is the reference's (n_v n_c / max degrees / vnode degrees / cnode degrees /
1-based adjacency lines, tab separated).  ``designs/ldpc/ieee80211n/1944.1296.txt`` is an
802.11n-style (1944,1296) QC code authored for this repo (the reference ships no such matrix).
"""
import ctypes

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as splg

from commpy_amd import _lib

__all__ = ['build_matrix', 'get_ldpc_code_params', 'ldpc_bp_decode', 'write_ldpc_params',
           'triang_ldpc_systematic_encode']

_llr_max = 500


def build_matrix(ldpc_code_params):
    """Add ``parity_check_matrix`` (CSC int8) and ``generator_matrix`` (CSR) to the dict -- ldpc.py:13-48.

    Like the reference the generator is ``inv(H[:, -n_c:]) . H[:, :-n_c]`` computed over the reals, valid
    only for (approximately) triangular systematic codes such as the WiMax designs (quirk B12).
    """
    n_cnodes = ldpc_code_params['n_cnodes']
    deg = ldpc_code_params['cnode_deg_list']
    adj = ldpc_code_params['cnode_adj_list'].reshape((n_cnodes, ldpc_code_params['max_cnode_deg']))
    rows = np.repeat(np.arange(n_cnodes), deg)
    cols = np.concatenate([adj[c, :deg[c]] for c in range(n_cnodes)])
    H = sp.coo_matrix((np.ones(len(rows), np.int8), (rows, cols)),
                      shape=(n_cnodes, ldpc_code_params['n_vnodes']))
    H.sum_duplicates()
    H.data[:] = 1
    H = H.astype(np.int8).tocsc()
    ldpc_code_params['parity_check_matrix'] = H
    ldpc_code_params['generator_matrix'] = splg.inv(H[:, -n_cnodes:]).dot(H[:, :-n_cnodes]).tocsr()


def get_ldpc_code_params(ldpc_design_filename, compute_matrix=False):
    """Parse a design file into the reference's parameter dictionary -- ldpc.py:51-141.

    Keys: n_vnodes, n_cnodes, max_vnode_deg, max_cnode_deg, vnode_adj_list, cnode_adj_list,
    vnode_cnode_map, cnode_vnode_map (int32, flattened, -1 padded, 0-based), vnode_deg_list,
    cnode_deg_list (int32) and, if asked, the matrices of ``build_matrix``.
    """
    with open(ldpc_design_filename) as f:
        n_vnodes, n_cnodes = [int(x) for x in f.readline().split(' ')]
        max_vnode_deg, max_cnode_deg = [int(x) for x in f.readline().split(' ')]
        vnode_deg_list = np.array([int(x) for x in f.readline().split(' ')[:-1]], np.int32)
        cnode_deg_list = np.array([int(x) for x in f.readline().split(' ')[:-1]], np.int32)
        vnode_adj = -np.ones([n_vnodes, max_vnode_deg], int)
        cnode_adj = -np.ones([n_cnodes, max_cnode_deg], int)
        for v in range(n_vnodes):
            vnode_adj[v, 0:vnode_deg_list[v]] = [int(x) - 1 for x in f.readline().split('\t')]
        for c in range(n_cnodes):
            cnode_adj[c, 0:cnode_deg_list[c]] = [int(x) - 1 for x in f.readline().split('\t')]

    # position of each node inside its neighbour's adjacency list (ldpc.py:112-121)
    cnode_vnode_map = -np.ones([n_cnodes, max_cnode_deg], int)
    vnode_cnode_map = -np.ones([n_vnodes, max_vnode_deg], int)
    for c in range(n_cnodes):
        for i, v in enumerate(cnode_adj[c, 0:cnode_deg_list[c]]):
            cnode_vnode_map[c, i] = np.where(vnode_adj[v, :] == c)[0][0]
    for v in range(n_vnodes):
        for i, c in enumerate(vnode_adj[v, 0:vnode_deg_list[v]]):
            vnode_cnode_map[v, i] = np.where(cnode_adj[c, :] == v)[0][0]

    params = {
        'n_vnodes': n_vnodes, 'n_cnodes': n_cnodes,
        'max_cnode_deg': max_cnode_deg, 'max_vnode_deg': max_vnode_deg,
        'cnode_adj_list': cnode_adj.flatten().astype(np.int32),
        'cnode_vnode_map': cnode_vnode_map.flatten().astype(np.int32),
        'vnode_adj_list': vnode_adj.flatten().astype(np.int32),
        'vnode_cnode_map': vnode_cnode_map.flatten().astype(np.int32),
        'cnode_deg_list': cnode_deg_list, 'vnode_deg_list': vnode_deg_list,
    }
    if compute_matrix:
        build_matrix(params)
    return params


def _edge_list(ldpc_code_params):
    """(edge_check, edge_var) int32 sorted by (check, variable): the row-major order SciPy keeps the
    reference's ``message_matrix`` in, taken from ``parity_check_matrix`` when present (ldpc.py:189-195)."""
    H = ldpc_code_params.get('parity_check_matrix')
    if H is None:
        build_matrix(ldpc_code_params)                       # the reference adds the matrices too (quirk B9)
        H = ldpc_code_params['parity_check_matrix']
    coo = sp.csr_matrix(H).tocoo()
    order = np.lexsort((coo.col, coo.row))
    keep = coo.data[order] != 0
    return _lib.as_i32(coo.row[order][keep]), _lib.as_i32(coo.col[order][keep])


def _device_code(ldpc_code_params):
    cache = ldpc_code_params.get('_cpx_ldpc')
    if cache is None:
        lib = _lib.load()
        _lib.require_device()
        ec, ev = _edge_list(ldpc_code_params)
        h = ctypes.c_void_p()
        _lib.check(lib.cpx_ldpc_create(int(ldpc_code_params['n_vnodes']), int(ldpc_code_params['n_cnodes']), len(ec),
                                       ec.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                       ev.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(h)))
        cache = _CodeHandle(h)
        ldpc_code_params['_cpx_ldpc'] = cache
    return cache.h


class _CodeHandle:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            _lib.load().cpx_ldpc_destroy(self.h)
        except Exception:
            pass


def ldpc_bp_decode(llr_vec, ldpc_code_params, decoder_algorithm, n_iters, return_iterations=False):
    """Belief-propagation LDPC decoding on MI355X; same signature/return as ldpc.py:144.

    ``llr_vec``: 1-D float array whose length is a multiple of n_vnodes (several blocks are decoded
    at once -- on the GPU they really are decoded in parallel); positive = bit 0.  It is clipped
    IN PLACE to [-500, 500] like the reference (when it is a float64 ndarray).  Returns
    ``(dec_word int8, out_llrs float64)``, each ``(n_vnodes, n_blocks)`` with one block per column,
    squeezed to 1-D for a single block.  ``return_iterations=True`` (extension) appends the int32
    number of executed iterations per block.
    """
    if decoder_algorithm not in ('SPA', 'MSA'):
        raise NameError('Please input a valid decoder_algorithm string (meanning "SPA" or "MSA").')
    lib = _lib.load()
    n_v = int(ldpc_code_params['n_vnodes'])
    llr = _lib.as_f64(llr_vec).reshape(-1)
    if llr.size % n_v:
        raise ValueError('llr_vec length must be a multiple of the block length')
    n_blocks = llr.size // n_v
    code = _device_code(ldpc_code_params)
    dec = np.zeros((n_v, n_blocks), dtype=np.int8)
    out = np.zeros((n_v, n_blocks))
    its = np.zeros(n_blocks, dtype=np.int32)
    if n_blocks:
        work = llr if llr.flags.writeable and llr.flags.c_contiguous else llr.copy()
        _lib.check(lib.cpx_ldpc_bp_decode_batch(code, _lib.ptr(work), n_blocks, 0 if decoder_algorithm == 'SPA' else 1,
                                                int(n_iters), _lib.ptr(dec), _lib.ptr(out), _lib.ptr(its)))
        if isinstance(llr_vec, np.ndarray) and llr_vec.dtype == np.float64 and work is not llr_vec:
            try:
                llr_vec[...] = work.reshape(llr_vec.shape)        # in-place clip (ldpc.py:186)
            except ValueError:
                pass
    dec, out = dec.squeeze(), out.squeeze()                        # (ldpc.py:251-253)
    return (dec, out, its) if return_iterations else (dec, out)


def write_ldpc_params(parity_check_matrix, file_path):
    """Write a parity-check matrix as a design file -- ldpc.py:257-299 (same text format, mode 'x')."""
    H = np.asarray(parity_check_matrix)
    with open(file_path, 'x') as f:
        f.write('{} {}\n'.format(H.shape[1], H.shape[0]))
        f.write('{} {}\n'.format(H.sum(0).max(), H.sum(1).max()))
        f.write(''.join('{} '.format(d) for d in H.sum(0)) + '\n')
        f.write(''.join('{} '.format(d) for d in H.sum(1)) + '\n')
        for line in H.T:
            f.write('\t'.join(str(node + 1) for node in line.nonzero()[0]) + '\n')
        for line in H:
            f.write('\t'.join(str(node + 1) for node in line.nonzero()[0]) + '\n')
        f.write('\n')


def triang_ldpc_systematic_encode(message_bits, ldpc_code_params, pad=True):
    """Systematic encoder for (approximately) triangular LDPC codes (host) -- ldpc.py:302-354."""
    if ldpc_code_params.get('generator_matrix') is None or ldpc_code_params.get('parity_check_matrix') is None:
        build_matrix(ldpc_code_params)
    block_length = ldpc_code_params['generator_matrix'].shape[1]
    modulo = len(message_bits) % block_length
    if modulo:
        if pad:
            message_bits = np.concatenate((message_bits, np.zeros(block_length - modulo, message_bits.dtype)))
        else:
            raise ValueError('Padding is disable but message length is not a multiple of block length.')
    message_bits = message_bits.reshape(block_length, -1, order='F')
    parity_part = ldpc_code_params['generator_matrix'].dot(message_bits) % 2
    return np.vstack((message_bits, parity_part)).squeeze().astype(np.int8)
