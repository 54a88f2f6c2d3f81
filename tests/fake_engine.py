"""A HOST stand-in for libcommpy_amd.so, for ONE purpose: walking bench.py's N > 1 control flow on a box without GPUs.

TEST INFRASTRUCTURE ONLY (tests/test_bench_launcher.py).  The product has no CPU fallback and this is not one: nothing in
commpy_amd/ can reach it, it serves exactly the C-ABI calls `bench.py --gpus N` makes as a rank, and it computes them with the
host mirrors of commpy_amd and the CPU oracle.  What it is for: `python bench.py --gpus 2` has never run with two ranks on
hardware (one GPU per lease, SCALE_r01..r04 skipped), so the rank path -- commpy_amd.parallel.RankComm's nonce rendezvous,
cpx_comm_init_rank -> cpx_comm_allgather_u8 -> cpx_comm_allreduce_* -> cpx_comm_info, the in-place gather layout, the checksum
exchange, the JSON fields `value_with_gather` / `gather` / `comm_world` -- is executed here end to end by two real processes
started by bench.py's own launcher.  "Device memory" is host memory (addresses of NumPy buffers); collectives go through files in
$CPX_FAKE_DIR keyed by a per-communicator sequence number.

    python tests/fake_engine.py <bench.py arguments>        (what bench.launch_ranks starts per rank in the test)
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

CALLS = []                                             # names of the entry points walked, in order (written to $CPX_FAKE_DIR at exit)


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    if isinstance(p, ctypes.Array):
        return ctypes.addressof(p)
    if hasattr(p, "value"):
        return int(p.value or 0)
    if hasattr(p, "_obj"):                             # ctypes.byref(x)
        return ctypes.addressof(p._obj)
    return ctypes.cast(p, ctypes.c_void_p).value or 0


def _view(p, nbytes, dtype=np.uint8):
    buf = (ctypes.c_char * int(nbytes)).from_address(_addr(p))
    return np.frombuffer(buf, dtype=dtype)


def _out(ref, value):
    """Store through a ctypes.byref(...) / POINTER argument."""
    ref._obj.value = value


class _Trellis:
    def __init__(self, k, n, S, I, nxt, out):
        self.k, self.n, self.number_states, self.number_inputs = k, n, S, I
        self.total_memory = int(round(np.log2(S)))
        self.next_state_table, self.output_table = nxt, out
        self.code_type = "default"


class FakeComm:
    def __init__(self, rank, world, tag):
        self.rank, self.world, self.tag, self.seq = rank, world, tag, 0
        self.dir = os.environ["CPX_FAKE_DIR"]

    def exchange(self, arr):
        """All ranks contribute `arr`; returns the list of all contributions in rank order."""
        self.seq += 1
        base = os.path.join(self.dir, "c%s_op%d" % (self.tag, self.seq))
        tmp = "%s_r%d.tmp.npy" % (base, self.rank)
        np.save(tmp, np.ascontiguousarray(arr))
        os.replace(tmp, "%s_r%d.npy" % (base, self.rank))
        out, deadline = [], time.time() + 120
        for r in range(self.world):
            path = "%s_r%d.npy" % (base, r)
            while not os.path.exists(path):
                if time.time() > deadline:
                    raise TimeoutError("fake collective: rank %d never arrived at %s" % (r, base))
                time.sleep(0.005)
            out.append(np.load(path))
        return out


class FakeLib:
    """The subset of include/commpy_amd.h that bench.py's rank path and commpy_amd.parallel.RankComm call."""

    def __init__(self):
        self.keep = {}                                 # address -> buffer (keeps "device" allocations alive)
        self.trellis, self.modem, self.comms, self.timers = {}, {}, {}, {}
        self.next_handle = 0x1000
        self.kernel = b""
        self.rank = int(os.environ.get("RANK", "0"))

    def __getattr__(self, name):                       # anything else is a bug in the test's assumptions: fail loudly
        raise AttributeError("tests/fake_engine.py does not implement %s" % name)

    def _handle(self, table, obj):
        self.next_handle += 0x10
        table[self.next_handle] = obj
        return self.next_handle

    # ---- runtime
    def cpx_last_error(self): return b"fake engine"
    def cpx_version(self): return 500
    def cpx_build_id(self): return b"full:fake;viterbi:fake"

    def cpx_device_count(self, n):
        _out(n, int(os.environ.get("CPX_FAKE_DEVICES", "2")))
        return 0

    def cpx_set_device(self, d): CALLS.append("cpx_set_device"); return 0
    def cpx_get_device(self, d): _out(d, self.rank); return 0
    def cpx_set_precision(self, m): return 0

    def cpx_last_kernel(self, buf, cap):
        ctypes.memmove(buf, self.kernel + b"\0", min(cap, len(self.kernel) + 1))
        return 0

    def cpx_malloc(self, ref, nbytes):
        a = np.zeros(max(int(nbytes), 8), dtype=np.uint8)
        self.keep[a.ctypes.data] = a
        _out(ref, a.ctypes.data)
        return 0

    def cpx_free(self, p):
        self.keep.pop(_addr(p), None)
        return 0

    def cpx_memcpy_h2d(self, dst, src, n): ctypes.memmove(_addr(dst), _addr(src), int(n)); return 0
    def cpx_memcpy_d2h(self, dst, src, n): ctypes.memmove(_addr(dst), _addr(src), int(n)); return 0
    def cpx_stream_sync(self, st): return 0

    def cpx_timer_create(self, ref): _out(ref, self._handle(self.timers, [0.0, 0.0])); return 0
    def cpx_timer_start(self, t, st): self.timers[_addr(t)][0] = time.perf_counter(); return 0
    def cpx_timer_stop(self, t, st): self.timers[_addr(t)][1] = time.perf_counter(); return 0
    def cpx_timer_elapsed_ms(self, t, ms): a, b = self.timers[_addr(t)]; _out(ms, max((b - a) * 1e3, 1e-3)); return 0
    def cpx_timer_destroy(self, t): self.timers.pop(_addr(t), None); return 0
    def cpx_sclk_probe_start(self, ref, ms): _out(ref, 1); return 0
    def cpx_sclk_probe_destroy(self, p): return 0
    def cpx_sclk_probe_read(self, p, mhz, iv):
        if mhz is not None: _out(mhz, 1.0)
        if iv is not None: _out(iv, 1.0)
        return 0

    # ---- handles
    def cpx_trellis_create(self, k, n, S, I, nxt, out, ref):
        nx = np.array(_view(nxt, 4 * S * I, np.int32)).reshape(S, I)
        ot = np.array(_view(out, 4 * S * I, np.int32)).reshape(S, I)
        _out(ref, self._handle(self.trellis, _Trellis(k, n, S, I, nx, ot)))
        return 0

    def cpx_trellis_destroy(self, h): return 0

    def cpx_modem_create(self, c, M, ref):
        _out(ref, self._handle(self.modem, np.array(_view(c, 16 * M, np.complex128))))
        return 0

    def cpx_modem_destroy(self, h): return 0

    # ---- link stages and the decoder (host mirrors / the CPU oracle)
    def cpx_random_bits_dev(self, d, n, seed, stream_id, st):
        _view(d, n)[:] = np.random.RandomState((int(seed) * 7919 + int(stream_id)) % (2 ** 31)).randint(0, 2, int(n))
        return 0

    def cpx_conv_encode_batch_dev(self, h, d_msg, B, nmsg, terminate, rsc, d_out, nout, st):
        from commpy_amd.channelcoding import conv_encode_batch
        tr = self.trellis[_addr(h)]
        msgs = np.array(_view(d_msg, B * nmsg)).reshape(B, nmsg)
        coded = conv_encode_batch(msgs, tr, "term" if terminate else "cont")
        assert coded.shape[1] == nout, (coded.shape, nout)
        _view(d_out, B * nout)[:] = coded.reshape(-1)
        return 0

    def cpx_modulate_dev(self, h, d_bits, nsym, d_sym, st):
        c = self.modem[_addr(h)]
        nb = int(np.log2(len(c)))
        bits = np.array(_view(d_bits, nsym * nb)).reshape(nsym, nb)
        lab = bits.dot(1 << np.arange(nb - 1, -1, -1))
        _view(d_sym, 16 * nsym, np.complex128)[:] = c[lab]
        return 0

    def cpx_awgn_dev(self, d_x, n, s_re, s_im, seed, stream_id, d_y, st):
        rs = np.random.RandomState((int(seed) * 104729 + int(stream_id)) % (2 ** 31))
        x = np.array(_view(d_x, 16 * n, np.complex128))
        _view(d_y, 16 * n, np.complex128)[:] = x + s_re * rs.randn(n) + 1j * s_im * rs.randn(n)
        return 0

    def cpx_demod_soft_dev(self, h, d_y, ns, n0, d_llr, st):
        import oracle
        c = self.modem[_addr(h)]
        nb = int(np.log2(len(c)))
        _view(d_llr, 8 * ns * nb, np.float64)[:] = oracle.demodulate(c, np.array(_view(d_y, 16 * ns, np.complex128)), "soft", n0)
        self.kernel = b"demod_soft (fake engine)"
        return 0

    def cpx_viterbi_set_path(self, m): return 0

    def cpx_viterbi_decode_batch_dev(self, h, d_in, B, length, L, T, tb, dtype, d_out, st):
        import oracle
        CALLS.append("cpx_viterbi_decode_batch_dev")
        x = np.array(_view(d_in, 8 * B * length, np.float64)).reshape(B, length)
        dec = oracle.viterbi_decode(x, self.trellis[_addr(h)], tb, ("hard", "soft", "unquantized")[dtype])
        _view(d_out, B * L)[:] = np.asarray(dec, dtype=np.uint8)[:, :L].reshape(-1)
        self.kernel = b"viterbi_cw_fused_kernel<fake engine>"
        return 0

    def cpx_count_errors_dev(self, d_a, sa, d_b, sb, B, nchunks, chunk, d_errs, st):
        a, b = np.array(_view(d_a, B * sa)).reshape(B, sa), np.array(_view(d_b, B * sb)).reshape(B, sb)
        n = nchunks * chunk
        e = (a[:, :n] ^ b[:, :n]).reshape(B, nchunks, chunk).sum(axis=2)
        _view(d_errs, 4 * B * nchunks, np.int32)[:] = e.reshape(-1)
        return 0

    # ---- collectives (what commpy_amd.parallel.RankComm calls)
    def cpx_comm_unique_id(self, buf):
        CALLS.append("cpx_comm_unique_id")
        ctypes.memmove(_addr(buf), os.urandom(128), 128)
        return 0

    def cpx_comm_init_rank(self, uid, world, rank, ref):
        CALLS.append("cpx_comm_init_rank")
        raw = bytes(_view(uid, 128)) if not isinstance(uid, (bytes, bytearray)) else bytes(uid)
        import hashlib
        _out(ref, self._handle(self.comms, FakeComm(int(rank), int(world), hashlib.sha1(raw).hexdigest()[:12])))
        return 0

    def cpx_comm_info(self, h, nr, nl, fr):
        CALLS.append("cpx_comm_info")
        c = self.comms[_addr(h)]
        _out(nr, c.world); _out(nl, 1); _out(fr, c.rank)
        return 0

    def cpx_comm_destroy(self, h):
        CALLS.append("cpx_comm_destroy")
        self.comms.pop(_addr(h), None)
        return 0

    def cpx_comm_allgather_u8(self, h, send, recv, nbytes, streams):
        CALLS.append("cpx_comm_allgather_u8")
        c = self.comms[_addr(h)]
        mine = np.array(_view(send[0], nbytes))          # (copied before the in-place receive buffer is written)
        parts = c.exchange(mine)
        _view(recv[0], nbytes * c.world)[:] = np.concatenate(parts)
        return 0

    def _allreduce(self, h, send, recv, count, op, dtype):
        c = self.comms[_addr(h)]
        parts = c.exchange(np.array(_view(send[0], 8 * count, dtype)))
        red = {0: np.sum, 1: np.max}.get(int(op))              # commpy_amd.parallel._OPS
        if red is None:
            raise ValueError("fake engine: reduction op %d" % op)
        _view(recv[0], 8 * count, dtype)[:] = red(np.stack(parts), axis=0)
        return 0

    def cpx_comm_allreduce_i64(self, h, send, recv, count, op, streams):
        CALLS.append("cpx_comm_allreduce_i64")
        return self._allreduce(h, send, recv, count, op, np.int64)

    def cpx_comm_allreduce_f64(self, h, send, recv, count, op, streams):
        CALLS.append("cpx_comm_allreduce_f64")
        return self._allreduce(h, send, recv, count, op, np.float64)


def install():
    from commpy_amd import _lib
    fake = FakeLib()
    _lib._lib = fake
    _lib.load = lambda: fake
    return fake


if __name__ == "__main__":
    import atexit
    import runpy
    install()

    def dump():
        d = os.environ.get("CPX_FAKE_DIR")
        if d:
            with open(os.path.join(d, "calls_rank%s.json" % os.environ.get("RANK", "x")), "w") as f:
                json.dump(CALLS, f)

    atexit.register(dump)
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
