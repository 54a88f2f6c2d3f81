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
import hashlib
import os
import tempfile
import zipfile

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as splg

from commpy_amd import _lib

__all__ = ['build_matrix', 'get_ldpc_code_params', 'ldpc_bp_decode', 'write_ldpc_params',
           'triang_ldpc_systematic_encode']

_llr_max = 500


def _cache_dir():
    """Directory of compiled designs: ``$CPX_CACHE_DIR`` or ``~/.cache/commpy_amd`` (``None`` when caching is off)."""
    d = os.environ.get('CPX_CACHE_DIR')
    if d == '':
        return None
    return d or os.path.join(os.path.expanduser('~'), '.cache', 'commpy_amd')


def _cache_load(name):
    d = _cache_dir()
    if d is None:
        return None
    try:
        with np.load(os.path.join(d, name), allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    except (OSError, ValueError, KeyError, zipfile.BadZipFile):
        return None


def _cache_store(name, arrays):
    d = _cache_dir()
    if d is None:
        return
    try:
        os.makedirs(d, exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=d, suffix='.tmp')
        with os.fdopen(fd, 'wb') as f:
            np.savez(f, **arrays)
        os.replace(tmp, os.path.join(d, name))                  # atomic: concurrent processes never see half a file
    except OSError:
        pass


def _H_from_adjacency(p):
    n_cnodes = p['n_cnodes']
    deg = np.asarray(p['cnode_deg_list'])
    adj = np.asarray(p['cnode_adj_list']).reshape((n_cnodes, p['max_cnode_deg']))
    rows = np.repeat(np.arange(n_cnodes), deg)
    cols = adj[np.arange(adj.shape[1])[None, :] < deg[:, None]]
    H = sp.coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(n_cnodes, p['n_vnodes']))
    H.sum_duplicates()
    H.data[:] = 1
    return H.astype(np.int8).tocsc()


def build_matrix(ldpc_code_params):
    """Add ``parity_check_matrix`` (CSC int8) and ``generator_matrix`` (CSR) to the dict -- ldpc.py:13-48.

    Like the reference the generator is ``inv(H[:, -n_c:]) . H[:, :-n_c]`` computed over the reals, valid
    only for (approximately) triangular systematic codes such as the WiMax designs (quirk B12).  For a dict that came
    from a design file the two matrices are kept in the compiled-design cache (keyed by the file's hash), so the
    sparse inversion runs once per design, not once per code object.
    """
    n_cnodes = ldpc_code_params['n_cnodes']
    sha = ldpc_code_params.get('_design_sha')
    H = _H_from_adjacency(ldpc_code_params)
    G = None
    z = _cache_load(sha + '.mat.npz') if sha else None
    if z is not None and tuple(z['G_shape']) == (n_cnodes, ldpc_code_params['n_vnodes'] - n_cnodes):
        G = sp.csr_matrix((z['G_data'], z['G_indices'], z['G_indptr']), shape=tuple(z['G_shape']))
    if G is None:
        G = splg.inv(H[:, -n_cnodes:]).dot(H[:, :-n_cnodes]).tocsr()
        if sha:
            _cache_store(sha + '.mat.npz', {'G_data': G.data, 'G_indices': G.indices, 'G_indptr': G.indptr,
                                            'G_shape': np.array(G.shape)})
    ldpc_code_params['parity_check_matrix'] = H
    ldpc_code_params['generator_matrix'] = G
    ldpc_code_params['_cpx_H'] = H                               # identity token: H is the design's own matrix


_PARAM_ARRAYS = ('cnode_adj_list', 'cnode_vnode_map', 'vnode_adj_list', 'vnode_cnode_map', 'cnode_deg_list',
                 'vnode_deg_list')


def _parse_design(text):
    """Design-file text -> parameter dictionary (ldpc.py:91-136), vectorised."""
    lines = text.split('\n')
    n_vnodes, n_cnodes = [int(x) for x in lines[0].split(' ')]
    max_vnode_deg, max_cnode_deg = [int(x) for x in lines[1].split(' ')]
    vnode_deg_list = np.array([int(x) for x in lines[2].split(' ')[:-1]], np.int32)
    cnode_deg_list = np.array([int(x) for x in lines[3].split(' ')[:-1]], np.int32)
    vnode_adj = -np.ones([n_vnodes, max_vnode_deg], int)
    cnode_adj = -np.ones([n_cnodes, max_cnode_deg], int)
    for v in range(n_vnodes):
        vnode_adj[v, 0:vnode_deg_list[v]] = [int(x) - 1 for x in lines[4 + v].split('\t')]
    for c in range(n_cnodes):
        cnode_adj[c, 0:cnode_deg_list[c]] = [int(x) - 1 for x in lines[4 + n_vnodes + c].split('\t')]

    # position of each node inside its neighbour's adjacency list (ldpc.py:112-121): first match, like np.where(..)[0][0]
    def positions(adj_a, deg_a, adj_b):
        out = -np.ones(adj_a.shape, int)
        rows, cols = np.nonzero(np.arange(adj_a.shape[1])[None, :] < deg_a[:, None])
        nb = adj_a[rows, cols]                                  # neighbour of (row, col)
        out[rows, cols] = np.argmax(adj_b[nb] == rows[:, None], axis=1)
        return out

    cnode_vnode_map = positions(cnode_adj, cnode_deg_list, vnode_adj)
    vnode_cnode_map = positions(vnode_adj, vnode_deg_list, cnode_adj)
    return {
        'n_vnodes': n_vnodes, 'n_cnodes': n_cnodes,
        'max_cnode_deg': max_cnode_deg, 'max_vnode_deg': max_vnode_deg,
        'cnode_adj_list': cnode_adj.flatten().astype(np.int32),
        'cnode_vnode_map': cnode_vnode_map.flatten().astype(np.int32),
        'vnode_adj_list': vnode_adj.flatten().astype(np.int32),
        'vnode_cnode_map': vnode_cnode_map.flatten().astype(np.int32),
        'cnode_deg_list': cnode_deg_list, 'vnode_deg_list': vnode_deg_list,
    }


def get_ldpc_code_params(ldpc_design_filename, compute_matrix=False):
    """Parse a design file into the reference's parameter dictionary -- ldpc.py:51-141.

    Keys: n_vnodes, n_cnodes, max_vnode_deg, max_cnode_deg, vnode_adj_list, cnode_adj_list,
    vnode_cnode_map, cnode_vnode_map (int32, flattened, -1 padded, 0-based), vnode_deg_list,
    cnode_deg_list (int32) and, if asked, the matrices of ``build_matrix``.

    Compiled-design cache: the file is hashed (SHA-256 of its bytes); the parsed arrays and the
    device blob of the Tanner graph (``cpx_ldpc_blob_build``: sorted edge list, row/column pointers, padded node tables)
    are stored under that hash in ``$CPX_CACHE_DIR`` (default ``~/.cache/commpy_amd``; empty string = off), so the text
    is parsed and the graph compiled once per design, and a decoder handle is created by uploading the blob.
    """
    with open(ldpc_design_filename, 'rb') as f:
        raw = f.read()
    sha = hashlib.sha256(raw).hexdigest()
    # the cache entry is keyed by the design file AND by the blob format this engine writes: an entry of another engine
    # version is simply not found, and a found one is validated (cpx_ldpc_blob_info: magic, sizes, checksum, every index) before
    # it is trusted -- a blob that fails is dropped and rebuilt, never handed to the decoder
    entry = '%s.v%d.npz' % (sha, _BLOB_FORMAT)
    z = _cache_load(entry)
    params = None
    if z is not None and all(k in z for k in _PARAM_ARRAYS + ('dims', 'blob')) and _blob_ok(z['blob']):
        n_v, n_c, mvd, mcd = [int(x) for x in z['dims']]
        params = {'n_vnodes': n_v, 'n_cnodes': n_c, 'max_cnode_deg': mcd, 'max_vnode_deg': mvd}
        for k in _PARAM_ARRAYS:
            params[k] = z[k]
        params['_cpx_blob'] = z['blob']
    if params is None:
        params = _parse_design(raw.decode())
        try:
            params['_cpx_blob'] = ldpc_design_blob(params)
            _cache_store(entry, dict({k: params[k] for k in _PARAM_ARRAYS}, blob=params['_cpx_blob'],
                                     dims=np.array([params['n_vnodes'], params['n_cnodes'],
                                                    params['max_vnode_deg'], params['max_cnode_deg']])))
        except (_lib.EngineError, ValueError):                   # library not built / code beyond an engine limit
            pass
    params['_design_sha'] = sha
    if compute_matrix:
        build_matrix(params)
    return params


_BLOB_FORMAT = 1        # LdpcBlobHeader.version of csrc/ldpc.hip; part of the cache entry's name


def _blob_ok(blob):
    """True when the engine accepts ``blob`` (host-only check, no GPU needed); False also when the library is missing."""
    try:
        b = np.ascontiguousarray(blob, dtype=np.uint8)
        return _lib.load().cpx_ldpc_blob_info(_lib.ptr(b), b.nbytes, None, None, None, None, None) == 0
    except (_lib.EngineError, OSError):
        return False


def _edges_from_adjacency(p):
    """(edge_check, edge_var) int32 sorted by (check, variable) straight from the adjacency lists -- the set the
    reference's lil-matrix assignment builds (ldpc.py:39-41: duplicates collapse, columns sorted)."""
    H = _H_from_adjacency(p).tocsr()
    H.sort_indices()
    return _lib.as_i32(np.repeat(np.arange(H.shape[0]), np.diff(H.indptr))), _lib.as_i32(H.indices)


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


def ldpc_design_blob(ldpc_code_params, edges=None):
    """Device blob of the code's Tanner graph (uint8 array; host-only work, no GPU needed): ``cpx_ldpc_blob_build`` over
    the edge list of the adjacency arrays (or ``edges = (edge_check, edge_var)``)."""
    lib = _lib.load()
    ec, ev = edges if edges is not None else _edges_from_adjacency(ldpc_code_params)
    n_v, n_c = int(ldpc_code_params['n_vnodes']), int(ldpc_code_params['n_cnodes'])
    i32p = ctypes.POINTER(ctypes.c_int32)
    need = ctypes.c_size_t(0)
    _lib.check(lib.cpx_ldpc_blob_build(n_v, n_c, len(ec), ec.ctypes.data_as(i32p), ev.ctypes.data_as(i32p), None, 0,
                                       ctypes.byref(need)))
    buf = np.zeros((need.value + 7) // 8, dtype=np.uint64)
    _lib.check(lib.cpx_ldpc_blob_build(n_v, n_c, len(ec), ec.ctypes.data_as(i32p), ev.ctypes.data_as(i32p), _lib.ptr(buf),
                                       buf.nbytes, ctypes.byref(need)))
    return buf.view(np.uint8)[:need.value].copy()


def _device_code(ldpc_code_params):
    """cpx_ldpc* of the current device (one per device, created on first use).  The reference adds the two matrices to
    the dict on first decode (ldpc.py:189-190, quirk B9) and so does this; the device tables come from the compiled
    blob when the dict still describes the design file it was read from, else from ``parity_check_matrix``."""
    if ldpc_code_params.get('parity_check_matrix') is None:
        build_matrix(ldpc_code_params)
    hs = ldpc_code_params.get('_cpx_ldpc')
    if hs is None:
        def create():
            H = ldpc_code_params.get('parity_check_matrix')
            blob = ldpc_code_params.get('_cpx_blob')
            if blob is None or H is not ldpc_code_params.get('_cpx_H'):
                blob = ldpc_design_blob(ldpc_code_params, _edge_list(ldpc_code_params))
            blob = np.ascontiguousarray(blob, dtype=np.uint8)
            h = ctypes.c_void_p()
            rc = _lib.load().cpx_ldpc_create_from_blob(_lib.ptr(blob), blob.nbytes, ctypes.byref(h))
            if rc == _lib.CPX_EINVAL and blob is not None and ldpc_code_params.get('_cpx_blob') is not None:
                # a cached blob the engine refuses (format change, damaged file): rebuild from the matrix instead of failing
                ldpc_code_params.pop('_cpx_blob', None)
                blob = np.ascontiguousarray(ldpc_design_blob(ldpc_code_params, _edge_list(ldpc_code_params)), dtype=np.uint8)
                rc = _lib.load().cpx_ldpc_create_from_blob(_lib.ptr(blob), blob.nbytes, ctypes.byref(h))
            _lib.check(rc)
            return h
        hs = ldpc_code_params['_cpx_ldpc'] = _lib.DeviceHandles(create, 'cpx_ldpc_destroy')
    return hs.get()


def ldpc_bp_decode(llr_vec, ldpc_code_params, decoder_algorithm, n_iters, return_iterations=False):
    """Belief-propagation LDPC decoding on MI355X; same signature/return as ldpc.py:144.

    ``llr_vec``: 1-D float array whose length is a multiple of n_vnodes (several blocks are decoded
    at once -- on the GPU they really are decoded in parallel); positive = bit 0.  It is clipped
    IN PLACE to [-500, 500] like the reference (when it is a float64 ndarray).  Returns
    ``(dec_word int8, out_llrs float64)``, each ``(n_vnodes, n_blocks)`` with one block per column,
    squeezed to 1-D for a single block.  ``return_iterations=True`` (extension) appends the int32
    number of executed iterations per block.

    Abnormal inputs behave like the reference's: +-inf is clipped to +-500 first; a NaN LLR propagates -- through 'SPA' by
    itself, through 'MSA' because NumPy's ``min`` / ``sign`` propagate it (ldpc.py:229-238): blocks with a NaN LLR are
    detected by the fast kernels and decoded a second time by a NaN-exact kernel.
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
    # block-major result buffers: the reference's results are F-ordered views of exactly this memory (ldpc.py:251-253,
    # `reshape(-1, n_blocks, order='F')`), so `.T` below returns its shape, dtype, values and strides without any transposition
    dec = np.zeros((n_blocks, n_v), dtype=np.int8)
    out = np.zeros((n_blocks, n_v))
    its = np.zeros(n_blocks, dtype=np.int32)
    if n_blocks:
        work = llr if llr.flags.writeable and llr.flags.c_contiguous else llr.copy()
        _lib.check(lib.cpx_ldpc_bp_decode_batch_bm(code, _lib.ptr(work), n_blocks, 0 if decoder_algorithm == 'SPA' else 1,
                                                   int(n_iters), _lib.ptr(dec), _lib.ptr(out), _lib.ptr(its)))
        if isinstance(llr_vec, np.ndarray) and llr_vec.dtype == np.float64 and work is not llr_vec:
            try:
                llr_vec[...] = work.reshape(llr_vec.shape)        # in-place clip (ldpc.py:186)
            except ValueError:
                pass
    dec, out = dec.T.squeeze(), out.T.squeeze()                    # (ldpc.py:251-253)
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
    """Systematic encoder for (approximately) lower-triangular LDPC codes (host input generator) -- what ldpc.py:302-354
    returns: int8 ``(n, n_blocks)`` with block b in COLUMN b, message part on top (squeezed to 1-D for one block).
    ``generator_matrix`` / ``parity_check_matrix`` are added to the dictionary when missing (``build_matrix``).  A message
    that does not fill its last block is zero-padded, or refused with ``ValueError`` when ``pad`` is false."""
    if any(ldpc_code_params.get(key) is None for key in ('generator_matrix', 'parity_check_matrix')):
        build_matrix(ldpc_code_params)
    gen = ldpc_code_params['generator_matrix']
    k = gen.shape[1]
    msg = np.asarray(message_bits)
    n_blocks, rest = divmod(len(msg), k)
    if rest:
        if not pad:
            raise ValueError('message of %d bits is not a whole number of %d-bit blocks and pad=False' % (len(msg), k))
        n_blocks += 1
        grown = np.zeros(n_blocks * k, dtype=msg.dtype)
        grown[:len(msg)] = msg
        msg = grown
    blocks = msg.reshape(n_blocks, k)                     # consecutive message bits fill one block (= one output column)
    parity = gen.dot(blocks.T) % 2                        # generator over the reals, reduced mod 2 like ldpc.py:353
    code = np.empty((k + parity.shape[0], n_blocks), dtype=np.int8)
    code[:k] = blocks.T
    code[k:] = parity
    return code.squeeze()
