"""Multi-GPU batch sharding of the decoders: contiguous codeword shards, RCCL collectives owned by the engine.

Codewords are independent, so the path shards with NO data-path collective: GPU g decodes the contiguous
slice ``shard_bounds(B, g, G)`` of the batch with its own copy of the code tables.  Two collectives exist around it,
both issued through the C-ABI (``cpx_comm_*``, csrc/comm.hip: RCCL over xGMI, loaded with dlopen) -- no torch:

* one all-gather of the decoded bits (uint8) when the caller wants the whole ``[B, L]`` result on every GPU -- the
  array ``viterbi_decode`` / ``ldpc_bp_decode`` return in the reference (convcode.py:749, ldpc.py:251-254);
* one all-reduce (sum, int64) of the error / bit counters of a Monte-Carlo sweep (links.py:252-260).

Two ways to span several GPUs:

``DeviceGroup(devices)``  ONE process drives all GPUs (``ncclCommInitAll``): the form a CommPy script uses --
    ``DeviceGroup().viterbi_decode(llr, trellis, None, 'soft')`` instead of ``viterbi_decode(...)``.  One host thread
    per device enqueues that device's work (ctypes drops the GIL, ``hipSetDevice`` is per thread).
``RankComm(rank, world)``  one process per GPU (``bench.py`` under ``torch.distributed.run``): rank 0 creates the
    128-byte RCCL id and hands it to the other ranks through a file in ``/tmp`` that carries a per-launch nonce
    (one node, as the bench contract says).

``sharded_decode`` / ``reduce_counters`` take any object with the small ``Collective`` protocol (``rank``, ``world``,
``allgather_rows``, ``allreduce``): ``RankComm`` on GPUs; the CPU tests (tests/test_parallel_gloo.py) plug a gloo-backed
stand-in into the same shard arithmetic.

The reference has no distributed code at all; this module is new work.
"""
import ctypes
import os
import tempfile
import time

import numpy as np

from commpy_amd import _lib

__all__ = ['shard_bounds', 'shard_counts', 'pad_shard', 'unpad_gathered', 'sharded_decode', 'reduce_counters',
           'exchange_unique_id', 'launch_nonce', 'RankComm', 'DeviceGroup']


# ---- shard arithmetic (pure host code) --------------------------------------------------------------------------------
def shard_counts(n_items, world_size):
    """Rows per rank: contiguous blocks, the first ``n_items % world_size`` ranks get one more."""
    base, rem = divmod(int(n_items), int(world_size))
    return [base + (1 if r < rem else 0) for r in range(world_size)]


def shard_bounds(n_items, rank, world_size):
    """``(start, stop)`` of rank's contiguous slice of a batch of ``n_items`` codewords."""
    counts = shard_counts(n_items, world_size)
    start = sum(counts[:rank])
    return start, start + counts[rank]


def pad_shard(local, n_total, rank, world):
    """Equal-size shards for one fused all-gather: ``local`` (rows of ``shard_bounds``) padded with zero rows to the
    largest shard.  Returns a C-contiguous array of ``max(shard_counts)`` rows."""
    counts = shard_counts(n_total, world)
    local = np.ascontiguousarray(local)
    if local.shape[0] != counts[rank]:
        raise ValueError('local shard has %d rows, expected %d' % (local.shape[0], counts[rank]))
    rows = max(counts)
    if local.shape[0] == rows:
        return local
    out = np.zeros((rows,) + local.shape[1:], dtype=local.dtype)
    out[:local.shape[0]] = local
    return out


def unpad_gathered(full, n_total, world):
    """Inverse of the padding: ``full`` is ``[world * max_rows, ...]`` rank-major; returns ``[n_total, ...]``."""
    counts = shard_counts(n_total, world)
    rows = max(counts) if counts else 0
    if len(set(counts)) == 1:
        return full[:n_total]
    return np.concatenate([full[r * rows:r * rows + counts[r]] for r in range(world)], axis=0)


def sharded_decode(decode_fn, batch_inputs, comm=None, n_total=None, gather=True):
    """Decode a batch sharded over the ranks of ``comm`` (one process per GPU).

    The decode path has no exchange step: with ``gather=False`` every rank returns only the rows
    ``shard_bounds(n_total, rank, world)`` it decoded (no collective at all).  ``gather=True`` reassembles the full
    result on every rank with ONE all-gather of equal-size padded shards.

    ``decode_fn(*shard_inputs) -> ndarray [rows, ...]`` is any of the batched decoders (e.g.
    ``lambda x: viterbi_decode(x, trellis, None, 'soft')``); ``batch_inputs`` are arrays whose first axis is the
    codeword index (every rank passes the same full arrays).  ``comm=None`` = a single process: the whole batch.
    """
    n_total = int(batch_inputs[0].shape[0]) if n_total is None else int(n_total)
    rank, world = (0, 1) if comm is None else (comm.rank, comm.world)
    lo, hi = shard_bounds(n_total, rank, world)
    local = np.asarray(decode_fn(*[a[lo:hi] for a in batch_inputs]))
    if not gather or world == 1:
        return local
    return comm.allgather_rows(local, n_total)


def reduce_counters(counters, comm=None):
    """Sum int64 counters (bit errors, bits sent per SNR point, links.py:252-260) over the ranks of ``comm``."""
    c = np.ascontiguousarray(counters, dtype=np.int64)
    if comm is None or comm.world == 1:
        return c.copy()
    return comm.allreduce(c, 'sum')


# ---- id exchange for one-process-per-GPU launches ------------------------------------------------------------------
_comm_seq = 0                      # communicators formed by this process so far: all ranks create them in the same order


def _launcher_start_ticks():
    """Start time of the parent process -- the launcher all ranks of one job share -- in clock ticks since boot (field 22
    of /proc/<pid>/stat: what the kernel recorded at fork, the same number for every reader and never the same for a later
    process that reuses the pid).  '0' if /proc is not there."""
    try:
        with open('/proc/%d/stat' % os.getppid(), 'rb') as f:
            stat = f.read().decode('ascii', 'replace')
        return stat[stat.rindex(')') + 2:].split()[19]          # fields after "pid (comm) ": state is field 3 -> index 19 = 22
    except (OSError, ValueError, IndexError):
        return '0'


def launch_nonce(seq=0):
    """What identifies ONE communicator of ONE launch, computed identically by all its ranks: ``$CPX_COMM_NONCE`` when the
    launcher set one (bench.py's own launcher does: random per launch), else launcher pid + its start ticks; plus
    ``MASTER_PORT``, the elastic run id / restart count and ``seq`` (communicators this process has formed before)."""
    env = os.environ
    base = env.get('CPX_COMM_NONCE') or 'pid%d.%s' % (os.getppid(), _launcher_start_ticks())
    return '%s.%s.%s.%s.%d' % (base, env.get('MASTER_PORT', '0'), env.get('TORCHELASTIC_RUN_ID', 'none'),
                               env.get('TORCHELASTIC_RESTART_COUNT', '0'), seq)


def _default_id_path(seq):
    import hashlib
    return os.path.join(tempfile.gettempdir(), 'cpx_comm_%s.id' % hashlib.sha256(launch_nonce(seq).encode()).hexdigest()[:24])


_ID_MAGIC = b'CPXID1\n'


def exchange_unique_id(rank, world, make_id, path=None, timeout=180.0, seq=0, nonce=None):
    """Rank 0 calls ``make_id() -> bytes`` and publishes the result; every rank returns the same bytes.

    Single-node launches only (the bench contract): the id travels through a file.  Its default name is a hash of
    ``launch_nonce(seq)``, and the file CONTENT starts with that nonce: a reader accepts a file only if it carries the nonce
    of its own launch, so a file left behind by another (crashed) job is never taken for this communicator's whatever its
    age -- no clock or mtime is consulted.  Rank 0 removes a leftover of the same name before it creates the id; the write
    is atomic (temporary file + rename)."""
    if world == 1:
        return make_id()
    if nonce is None:
        nonce = launch_nonce(seq)
    if path is None:
        path = _default_id_path(seq)
    head = _ID_MAGIC + nonce.encode() + b'\n'
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
        blob = make_id()
        fd, tmp = tempfile.mkstemp(dir=os.path.dirname(path) or '.')
        with os.fdopen(fd, 'wb') as f:
            f.write(head + blob)
        os.replace(tmp, path)
        return blob
    deadline = time.time() + timeout
    seen_foreign = False
    while True:
        try:
            with open(path, 'rb') as f:
                data = f.read()
            if data.startswith(head) and len(data) > len(head):
                return data[len(head):]
            seen_foreign = seen_foreign or bool(data)
        except OSError:
            pass
        if time.time() > deadline:
            raise TimeoutError('rank %d: no communicator id for launch %r at %s after %.0f s%s' % (
                rank, nonce, path, timeout, ' (a file of another launch is there)' if seen_foreign else ''))
        time.sleep(0.01)


def _ptrs(values):
    arr = (ctypes.c_void_p * len(values))()
    for i, v in enumerate(values):
        arr[i] = v.value if isinstance(v, ctypes.c_void_p) else v
    return arr


_OPS = {'sum': 0, 'max': 1}


class RankComm:
    """RCCL communicator of a one-process-per-GPU job: this process is ``rank`` of ``world`` and owns ``device``."""

    def __init__(self, rank, world, device=None, id_path=None, timeout=180.0):
        self.lib = _lib.load()
        _lib.require_device()
        self.rank, self.world = int(rank), int(world)
        if device is not None:
            _lib.check(self.lib.cpx_set_device(int(device)))
        self.device = _lib.current_device()

        def make_id():
            buf = ctypes.create_string_buffer(128)
            _lib.check(self.lib.cpx_comm_unique_id(buf))
            return buf.raw

        global _comm_seq
        seq, _comm_seq = _comm_seq, _comm_seq + 1
        uid = exchange_unique_id(self.rank, self.world, make_id, id_path, timeout, seq)
        self._id_path = id_path if id_path is not None else _default_id_path(seq)
        self._own_path = id_path is None
        self.h = ctypes.c_void_p()
        _lib.check(self.lib.cpx_comm_init_rank(uid, self.world, self.rank, ctypes.byref(self.h)))

    # -- device-pointer collectives (asynchronous on `stream`) -----------------------------------------------------
    def allgather_dev(self, d_send, d_recv, bytes_per_rank, stream=None):
        _lib.check(self.lib.cpx_comm_allgather_u8(self.h, _ptrs([d_send]), _ptrs([d_recv]), int(bytes_per_rank),
                                                  _ptrs([stream]) if stream else None))

    def allreduce_dev(self, d_send, d_recv, count, dtype='i64', op='sum', stream=None):
        fn = self.lib.cpx_comm_allreduce_i64 if dtype == 'i64' else self.lib.cpx_comm_allreduce_f64
        _lib.check(fn(self.h, _ptrs([d_send]), _ptrs([d_recv]), int(count), _OPS[op], _ptrs([stream]) if stream else None))

    # -- host-array collectives (the Collective protocol of sharded_decode / reduce_counters) -------------------------
    def allgather_rows(self, local, n_total):
        from commpy_amd.devicelink import DeviceBuf
        local = np.ascontiguousarray(local)
        padded = pad_shard(local, n_total, self.rank, self.world)
        nb = padded.nbytes
        d_full = DeviceBuf(nb * self.world)
        mine = ctypes.c_void_p(d_full.ptr.value + self.rank * nb)
        _lib.check(self.lib.cpx_memcpy_h2d(mine, _lib.ptr(padded), nb))
        self.allgather_dev(mine, d_full.ptr, nb)                      # in place
        _lib.check(self.lib.cpx_stream_sync(None))
        full = d_full.to_array((self.world * padded.shape[0],) + padded.shape[1:], padded.dtype)
        return unpad_gathered(full, n_total, self.world)

    def allreduce(self, arr, op='sum'):
        from commpy_amd.devicelink import DeviceBuf
        a = np.ascontiguousarray(arr)
        if a.dtype not in (np.int64, np.float64):
            raise TypeError('allreduce: int64 or float64 arrays')
        d = DeviceBuf.from_array(a)
        self.allreduce_dev(d.ptr, d.ptr, a.size, 'i64' if a.dtype == np.int64 else 'f64', op)
        _lib.check(self.lib.cpx_stream_sync(None))
        return d.to_array(a.shape, a.dtype)

    def barrier(self):
        self.allreduce(np.zeros(1, np.int64))

    def close(self):
        if getattr(self, 'h', None):
            self.lib.cpx_comm_destroy(self.h)
            self.h = ctypes.c_void_p()
            if self.rank == 0 and self.world > 1 and self._own_path:
                try:
                    os.remove(self._id_path)
                except OSError:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceGroup:
    """Several GPUs driven by ONE process: shards a batch over ``devices`` and reassembles the result.

    ``devices``: list of device ordinals (default: all visible).  Code tables are replicated (one engine handle per
    device, created on first use); every device has its own stream; collectives go through one RCCL communicator
    (``ncclCommInitAll``).  Methods mirror the reference's function signatures.
    """

    def __init__(self, devices=None):
        self.lib = _lib.load()
        _lib.require_device()
        n = _lib.device_count()
        self.devices = list(range(n)) if devices is None else [int(d) for d in devices]
        if not self.devices:
            raise ValueError('DeviceGroup: no devices')
        self.G = len(self.devices)
        self._home = _lib.current_device()
        self.streams = []
        for d in self.devices:
            _lib.check(self.lib.cpx_set_device(d))
            s = ctypes.c_void_p()
            _lib.check(self.lib.cpx_stream_create(ctypes.byref(s)))
            self.streams.append(s)
        _lib.check(self.lib.cpx_set_device(self._home))
        self.h = ctypes.c_void_p()                    # RCCL communicator: formed by the first collective (`_comm`)
        self._closed = False

    def _comm(self):
        """The group's RCCL communicator (ncclCommInitAll), created on first use: a group that never gathers or reduces --
        one device, or gather=False -- does not need librccl at all."""
        if not self.h:
            devs = (ctypes.c_int * self.G)(*self.devices)
            _lib.check(self.lib.cpx_comm_init_all(devs, self.G, ctypes.byref(self.h)))
        return self.h

    # -- plumbing ----------------------------------------------------------------------------------------------------
    def each(self, fn):
        """Run ``fn(i, device, stream)`` once per device, each on its own host thread with its device current; returns
        the results in device order (exceptions propagate)."""
        import concurrent.futures as cf

        def work(i):
            _lib.check(self.lib.cpx_set_device(self.devices[i]))
            return fn(i, self.devices[i], self.streams[i])

        if self.G == 1:
            try:
                return [work(0)]
            finally:
                _lib.check(self.lib.cpx_set_device(self._home))
        with cf.ThreadPoolExecutor(self.G) as ex:
            return list(ex.map(work, range(self.G)))

    def prepare(self, fn):
        """Call ``fn()`` once per device, serially, with that device current: creates the per-device engine handles of a
        shared host object (Trellis, Modem, LDPC dict) before the per-device threads use them concurrently."""
        try:
            for d in self.devices:
                _lib.check(self.lib.cpx_set_device(d))
                fn()
        finally:
            _lib.check(self.lib.cpx_set_device(self._home))

    def sync(self):
        for d, s in zip(self.devices, self.streams):
            _lib.check(self.lib.cpx_set_device(d))
            _lib.check(self.lib.cpx_stream_sync(s))
        _lib.check(self.lib.cpx_set_device(self._home))

    def allgather_dev(self, d_send, d_recv, bytes_per_rank):
        """``d_send[i]`` / ``d_recv[i]``: device pointers on device i; asynchronous on the group's streams."""
        _lib.check(self.lib.cpx_comm_allgather_u8(self._comm(), _ptrs(d_send), _ptrs(d_recv), int(bytes_per_rank),
                                                  _ptrs(self.streams)))

    def allreduce_dev(self, d_send, d_recv, count, dtype='i64', op='sum'):
        fn = self.lib.cpx_comm_allreduce_i64 if dtype == 'i64' else self.lib.cpx_comm_allreduce_f64
        _lib.check(fn(self._comm(), _ptrs(d_send), _ptrs(d_recv), int(count), _OPS[op], _ptrs(self.streams)))

    def allreduce_counters(self, per_device_counters):
        """Sum one int64 counter array per device over the group with an RCCL all-reduce; returns the total (read back
        from the first device)."""
        from commpy_amd.devicelink import DeviceBuf
        arrs = [np.ascontiguousarray(c, dtype=np.int64) for c in per_device_counters]
        if len(arrs) != self.G or any(a.shape != arrs[0].shape for a in arrs):
            raise ValueError('allreduce_counters: one equally shaped array per device')
        if self.G == 1:
            return arrs[0].copy()                      # nothing to reduce: no communicator, no device round trip
        bufs = self.each(lambda i, dev, st: DeviceBuf.from_array(arrs[i]))
        self.allreduce_dev([b.ptr for b in bufs], [b.ptr for b in bufs], arrs[0].size)
        self.sync()
        out = self.each(lambda i, dev, st: bufs[i].to_array(arrs[0].shape, np.int64) if i == 0 else None)[0]
        for b in bufs:
            b.free()
        return out

    # -- sharded decoders ----------------------------------------------------------------------------------------------
    def viterbi_decode(self, coded_bits, trellis, tb_depth=None, decoding_type='hard', gather=True):
        """``viterbi_decode`` (convcode.py:661) of a batch ``[B, len]`` sharded over the group; int64 ``[B, L]``.

        ``gather=True``: the decoded bits are all-gathered over xGMI so that every GPU holds the whole result (read
        back from the first); ``gather=False``: no collective, every shard comes back from its own GPU."""
        from commpy_amd.channelcoding.convcode import _VIT_TYPES, _viterbi_sizes
        from commpy_amd.devicelink import DeviceBuf
        if decoding_type not in _VIT_TYPES:
            raise ValueError('The available decoding types are "hard", "soft" and "unquantized')
        x = _lib.as_f64(np.atleast_2d(coded_bits))
        B, length = x.shape
        L, n_steps, tb = _viterbi_sizes(length, trellis, tb_depth)
        if B == 0 or L == 0:
            return np.zeros((B, L), dtype=np.int64)
        if tb < 2:
            raise ValueError('tb_depth must be >= 2')
        counts = shard_counts(B, self.G)
        rows = max(counts)
        lib = self.lib
        self.prepare(trellis._device_handle)

        def launch(i, dev, st):
            lo, hi = shard_bounds(B, i, self.G)
            cnt = hi - lo
            d_in = DeviceBuf(max(cnt, 1) * length * 8)
            d_out = DeviceBuf((self.G if gather else 1) * rows * L)
            mine = ctypes.c_void_p(d_out.ptr.value + (i * rows * L if gather else 0))
            if cnt:
                _lib.check(lib.cpx_memcpy_h2d_async(d_in.ptr, _lib.ptr(x[lo:hi]), cnt * length * 8, st))
                _lib.check(lib.cpx_viterbi_decode_batch_dev(trellis._device_handle(), d_in.ptr, cnt, length, L, n_steps,
                                                            tb, _VIT_TYPES[decoding_type], mine, st))
            return d_in, d_out, mine

        bufs = self.each(launch)
        if gather and self.G > 1:
            self.allgather_dev([b[2] for b in bufs], [b[1].ptr for b in bufs], rows * L)
        self.sync()
        if gather:
            full = self.each(lambda i, dev, st: bufs[i][1].to_array((self.G * rows, L), np.uint8) if i == 0 else None)[0]
            out = unpad_gathered(full, B, self.G)
        else:
            parts = self.each(lambda i, dev, st: bufs[i][1].to_array((rows, L), np.uint8)[:counts[i]])
            out = np.concatenate(parts, axis=0)
        for b in bufs:
            b[0].free(); b[1].free()
        return out.astype(np.int64)

    def ldpc_bp_decode(self, llr_vec, ldpc_code_params, decoder_algorithm, n_iters, gather=True):
        """``ldpc_bp_decode`` (ldpc.py:144) of ``B`` blocks sharded over the group (BASELINE config 4: 262144 blocks ->
        32768 per GPU).  Returns ``(dec_word int8 (n, B), out_llrs float64 (n, B))`` like the reference; ``dec_word`` is
        all-gathered over xGMI (``gather=True``), ``out_llrs`` (8 bytes per bit) stays sharded and is read back from each GPU."""
        from commpy_amd.channelcoding.ldpc import _device_code
        from commpy_amd.devicelink import DeviceBuf
        if decoder_algorithm not in ('SPA', 'MSA'):
            raise NameError('Please input a valid decoder_algorithm string (meanning "SPA" or "MSA").')
        n_v = int(ldpc_code_params['n_vnodes'])
        if isinstance(llr_vec, np.ndarray) and llr_vec.dtype == np.float64:
            np.clip(llr_vec, -500, 500, out=llr_vec)                  # the reference clips the caller's array (ldpc.py:186)
        llr = _lib.as_f64(llr_vec).reshape(-1)
        if llr.size % n_v:
            raise ValueError('llr_vec length must be a multiple of the block length')
        B = llr.size // n_v
        if B == 0:
            return np.zeros((n_v, 0), np.int8).squeeze(), np.zeros((n_v, 0)).squeeze()
        if ldpc_code_params.get('parity_check_matrix') is None:
            from commpy_amd.channelcoding.ldpc import build_matrix
            build_matrix(ldpc_code_params)                            # once, before the per-device threads start
        llr = llr.reshape(B, n_v)
        counts = shard_counts(B, self.G)
        rows = max(counts)
        alg = 0 if decoder_algorithm == 'SPA' else 1
        lib = self.lib
        self.prepare(lambda: _device_code(ldpc_code_params))

        def launch(i, dev, st):
            lo, hi = shard_bounds(B, i, self.G)
            cnt = hi - lo
            d_llr, d_out = DeviceBuf(max(cnt, 1) * n_v * 8), DeviceBuf(max(cnt, 1) * n_v * 8)
            d_dec = DeviceBuf((self.G if gather else 1) * rows * n_v)
            mine = ctypes.c_void_p(d_dec.ptr.value + (i * rows * n_v if gather else 0))
            if cnt:
                _lib.check(lib.cpx_memcpy_h2d_async(d_llr.ptr, _lib.ptr(llr[lo:hi]), cnt * n_v * 8, st))
                # block-major outputs: a shard is `cnt` contiguous rows, the gathered array is [B][n_v] = the memory of the
                # reference's F-ordered result (ldpc.py:251-253); no transposition on the device
                _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(_device_code(ldpc_code_params), d_llr.ptr, cnt, alg, int(n_iters),
                                                               mine, d_out.ptr, None, st))
            return d_llr, d_out, d_dec, mine

        bufs = self.each(launch)
        if gather and self.G > 1:
            self.allgather_dev([b[3] for b in bufs], [b[2].ptr for b in bufs], rows * n_v)
        self.sync()
        outs = self.each(lambda i, dev, st: bufs[i][1].to_array((counts[i], n_v), np.float64) if counts[i] else np.zeros((0, n_v)))
        if gather:
            raw = self.each(lambda i, dev, st: bufs[i][2].to_array((self.G, rows, n_v), np.int8) if i == 0 else None)[0]
            decs = [raw[g, :counts[g]] for g in range(self.G)]
        else:
            decs = self.each(lambda i, dev, st: bufs[i][2].to_array((rows, n_v), np.int8)[:counts[i]])
        for b in bufs:
            for d in b[:3]:
                d.free()
        return np.concatenate(decs, axis=0).T.squeeze(), np.concatenate(outs, axis=0).T.squeeze()

    def wifi_ber_sweep(self, mcs, snrs_db, n_bits, send_chunk=600, frame_aggregation=1, generator_matrix=None, seed=1):
        """BASELINE config 5: BER of an 802.11 MCS over AWGN per SNR point, ``n_bits`` information bits per point split
        over the group.  Every GPU simulates its share with ``DeviceWifiLink`` (own Philox seed); the int64 error and bit
        counters are summed with an RCCL all-reduce (links.py:252-260).  Returns ``(ber, bit_errors, bits)`` per SNR."""
        from commpy_amd.devicelink import DeviceWifiLink
        import math
        share = int(math.ceil(n_bits / self.G))

        def run(i, dev, st):
            link = DeviceWifiLink(mcs, send_chunk, frame_aggregation, generator_matrix, seed=seed + 7919 * i)
            ber = link.ber_sweep_batched(snrs_db, share)
            T = int(math.ceil(share / link.nbits))
            bits = T * link.nbits
            return np.stack([np.rint(ber * bits).astype(np.int64), np.full(len(snrs_db), bits, np.int64)])

        total = self.allreduce_counters(self.each(run))
        return total[0] / total[1].astype(float), total[0], total[1]

    def close(self):
        if getattr(self, '_closed', True):
            return
        self._closed = True
        if self.h:
            self.lib.cpx_comm_destroy(self.h)
            self.h = ctypes.c_void_p()
        for d, s in zip(self.devices, self.streams):
            self.lib.cpx_set_device(d)
            self.lib.cpx_stream_destroy(s)
        self.lib.cpx_set_device(self._home)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
