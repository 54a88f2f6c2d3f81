"""Round-2 additions on the GPU: fused hard-demod -> hard-Viterbi entry point, compiled LDPC designs, the engine's RCCL
collectives (communicators of one device: the collectives really run through librccl, trivially), single-process
DeviceGroup drivers, handle/device checks, kernel-name reporting."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from helpers import golden, ldpc_params, make_trellis

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ fused hard demodulation + Viterbi
def _modems():
    from commpy_amd.modulation import Modem, PSKModem, QAMModem
    return {"psk2": PSKModem(2), "qam4": QAMModem(4), "psk8": PSKModem(8), "qam16": QAMModem(16), "qam64": QAMModem(64),
            "custom4": Modem([1 + 1j, -1.2 + 0.8j, 0.3 - 1j, -1 - 1.5j])}


@pytest.mark.parametrize("tname", ["t57", "k7_133_171", "k2_default", "rsc_legacy_8", "r13_k4"])
def test_fused_hard_demod_viterbi_equals_two_calls(gpu, tname):
    from commpy_amd import _lib
    from commpy_amd.channelcoding import conv_encode_batch, viterbi_decode
    tr = make_trellis(tname)
    rs = np.random.RandomState(len(tname))
    for mname, md in _modems().items():
        nb = md.num_bits_symbol
        B, nmsg = 37, 60 * tr.k
        coded = conv_encode_batch(rs.randint(0, 2, (B, nmsg)), tr)
        pad = (-coded.shape[1]) % nb                                   # whole symbols
        coded = np.concatenate([coded, np.zeros((B, pad), coded.dtype)], axis=1)
        s = md.modulate(coded.reshape(-1)).reshape(B, -1)
        N0 = md.Es / 10 ** 0.9
        y = s + np.sqrt(N0 / 2) * (rs.randn(*s.shape) + 1j * rs.randn(*s.shape))
        bits = md.demodulate(y.reshape(-1), "hard").reshape(B, -1)
        for tb in (None, 12):
            two = viterbi_decode(bits, tr, tb, "hard")
            one = md.demodulate_viterbi_hard(y, tr, tb)
            assert "demod" in _lib.last_kernel(), _lib.last_kernel()
            assert one.dtype == np.int64 and np.array_equal(one, two), (tname, mname, tb)
        want = oracle.viterbi_decode(oracle.demodulate(md.constellation, y[3], "hard"), tr, None, "hard")
        assert np.array_equal(md.demodulate_viterbi_hard(y[3], tr), want)          # 1-D in -> 1-D out, vs the oracle pair


def test_fused_hard_demod_viterbi_boundaries_and_limits(gpu):
    """Symbols exactly on decision boundaries take the first-minimum label in both paths; 128-state trellises take the
    two-call path inside the wrapper."""
    from commpy_amd.channelcoding import viterbi_decode
    from commpy_amd.modulation import QAMModem
    md = QAMModem(16)
    tr = make_trellis("k7_133_171")
    rs = np.random.RandomState(1)
    y = (rs.randint(-4, 5, (9, 64)) + 1j * rs.randint(-4, 5, (9, 64))).astype(complex)   # grid lines and midpoints
    bits = md.demodulate(y.reshape(-1), "hard").reshape(9, -1)
    assert np.array_equal(md.demodulate_viterbi_hard(y, tr), viterbi_decode(bits, tr, None, "hard"))
    big = make_trellis("k8_247_371")
    y2 = y[:, :32]
    bits2 = md.demodulate(y2.reshape(-1), "hard").reshape(9, -1)
    assert np.array_equal(md.demodulate_viterbi_hard(y2, big), viterbi_decode(bits2, big, None, "hard"))
    assert md.demodulate_viterbi_hard(np.zeros((0, 8), complex), tr).shape == (0, 16)


# ------------------------------------------------------------------ compiled LDPC designs
def test_ldpc_handle_from_blob_decodes_like_edge_list(gpu, tmp_path, monkeypatch):
    from commpy_amd.channelcoding import ldpc as L
    monkeypatch.setenv("CPX_CACHE_DIR", str(tmp_path))
    design = os.path.join(os.path.dirname(L.__file__), "designs/ldpc/ieee80211n/1944.1296.txt")
    g = golden("ldpc_c4x")
    llr = g["e9__llr"][:1944 * 4]
    p1 = L.get_ldpc_code_params(design)              # cold: parses, compiles, stores
    p2 = L.get_ldpc_code_params(design)              # warm: arrays + blob from the cache
    p3 = ldpc_params("n1944")                        # no blob: edge list from the matrices
    outs = [L.ldpc_bp_decode(llr.copy(), p, "MSA", 50) for p in (p1, p2, p3)]
    for d, o in outs[1:]:
        assert np.array_equal(d, outs[0][0]) and np.array_equal(o, outs[0][1])
    assert np.array_equal(outs[0][0], g["e9__dec_MSA"][:4].T)
    assert "parity_check_matrix" in p1 and "generator_matrix" in p1                 # quirk B9 kept


def test_handles_are_per_device_and_checked(gpu):
    """A handle used while another device is current is an error, not a fault; the host objects create one handle per
    device on demand."""
    from commpy_amd import _lib
    lib = _lib.load()
    tr = make_trellis("t57")
    h = tr._device_handle()
    assert tr._device_handle().value == h.value                                   # cached per device
    if _lib.device_count() < 2:
        return
    _lib.check(lib.cpx_set_device(1))
    try:
        d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.cpx_malloc(ctypes.byref(d_in), 8 * 20))
        _lib.check(lib.cpx_malloc(ctypes.byref(d_out), 10))
        assert lib.cpx_viterbi_decode_batch_dev(h, d_in, 1, 20, 10, 11, 10, 0, d_out, None) == _lib.CPX_EINVAL
        assert "device" in _lib.last_error()
        assert tr._device_handle().value != h.value
    finally:
        _lib.check(lib.cpx_set_device(0))


# ------------------------------------------------------------------ collectives and multi-GPU drivers
def test_rank_comm_world1_collectives(gpu):
    """cpx_comm_init_rank / allgather / allreduce through librccl with a communicator of one rank."""
    from commpy_amd.parallel import RankComm, reduce_counters, sharded_decode
    comm = RankComm(0, 1)
    rows = np.arange(35, dtype=np.uint8).reshape(7, 5)
    assert np.array_equal(comm.allgather_rows(rows, 7), rows)
    assert np.array_equal(comm.allreduce(np.array([5, -2, 1 << 40], np.int64)), [5, -2, 1 << 40])
    assert np.array_equal(comm.allreduce(np.array([0.5, 2.0]), "max"), [0.5, 2.0])
    comm.barrier()
    assert np.array_equal(reduce_counters([1, 2, 3], comm), [1, 2, 3])
    assert np.array_equal(sharded_decode(lambda x: x * 2, [rows], comm), rows * 2)
    comm.close()


def test_device_group_viterbi_ldpc_and_counters(gpu):
    """DeviceGroup over every visible GPU (one on the test box): sharded Viterbi and LDPC decodes equal the
    single-device calls, gathered or not; counters are all-reduced through RCCL."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding import conv_encode_batch, ldpc_bp_decode, viterbi_decode
    from commpy_amd.parallel import DeviceGroup
    grp = DeviceGroup()
    assert grp.G == _lib.device_count()
    tr = make_trellis("k7_133_171")
    rs = np.random.RandomState(2)
    msgs = rs.randint(0, 2, (203, 128))
    llr = 6.0 * (2.0 * conv_encode_batch(msgs, tr) - 1) + 4.0 * rs.randn(203, 268)
    want = viterbi_decode(llr, tr, None, "soft")
    for gather in (True, False):
        got = grp.viterbi_decode(llr, tr, None, "soft", gather=gather)
        assert got.dtype == np.int64 and np.array_equal(got, want), gather
    g = golden("ldpc_c4x")
    p = ldpc_params("n1944")
    x = g["e9__llr"].copy()
    dec, out = grp.ldpc_bp_decode(x, p, "MSA", 50)
    d1, o1 = ldpc_bp_decode(g["e9__llr"].copy(), p, "MSA", 50)
    assert np.array_equal(dec, d1) and np.array_equal(out, o1) and np.array_equal(dec, g["e9__dec_MSA"].T)
    tot = grp.allreduce_counters([np.array([3 + i, 10], np.int64) for i in range(grp.G)])
    assert tot[1] == 10 * grp.G and tot[0] == sum(3 + i for i in range(grp.G))
    ber, errs, bits = grp.wifi_ber_sweep(5, np.array([14.0, 40.0]), 600 * 400, generator_matrix=[[0o133, 0o171]])
    assert bits[0] >= 600 * 400 and errs[1] == 0 and 0 < ber[0] < 0.5
    grp.close()


# ------------------------------------------------------------------ ADVICE items that need the device
def test_conv_encode_gpu_rsc_termination_string(gpu):
    from commpy_amd.channelcoding import conv_encode
    from commpy_amd.devicelink import conv_encode_gpu
    rs = np.random.RandomState(5)
    msgs = rs.randint(0, 2, (9, 50))
    for tname in ("rsc_legacy_4", "rsc_legacy_8", "t57", "k7_133_171"):
        tr = make_trellis(tname)
        for term in ("term", "cont", "rsc"):
            got = conv_encode_gpu(msgs, tr, term)
            for b in (0, 8):
                want = conv_encode(msgs[b], tr, term)
                assert got[b].shape == want.shape and np.array_equal(got[b], want), (tname, term)


def test_wifi_custom_receiver_end_to_end(gpu):
    """Wifi80211.link_performance(receiver=...) with a per-transmission receiver: runs, one call per
    transmission, BER of a sensible receiver is sensible."""
    from commpy_amd.channels import SISOFlatChannel
    from commpy_amd.wifi80211 import Wifi80211
    w = Wifi80211(1, generator_matrix=[[0o133, 0o171]])
    calls = []

    def receiver(y, h, constellation, noise_var):
        calls.append(np.shape(y))
        return w.modem.demodulate(y, "soft", noise_var)

    np.random.seed(4)
    bers, bes, ces, ncs = w.link_performance(SISOFlatChannel(fading_param=(1 + 0j, 0j)), np.array([9.0]), 6, 10 ** 9, 240,
                                             receiver=receiver, stop_on_surpass_error=False)
    assert len(calls) == 6 and all(len(c) == 1 for c in calls)
    assert bers[0] < 0.05


def test_demod_soft_scaled_is_the_sign_flip_fused(gpu):
    """cpx_demod_soft_scaled_dev(scale = -1) == -cpx_demod_soft_dev bit for bit (the config-4 chain's sign flip, test_ldpc.py:53-54,
    without the extra pass); scale = 1 is the plain entry point; subnormal / zero noise_var keeps the division form."""
    import ctypes
    from commpy_amd import _lib
    from commpy_amd.devicelink import DeviceBuf
    from commpy_amd.modulation import PSKModem, QAMModem
    lib = _lib.load()
    rs = np.random.RandomState(4)
    for md in (QAMModem(64), QAMModem(4), PSKModem(8)):
        ns, nb = 5000, md.num_bits_symbol
        y = md.constellation[rs.randint(0, md.m, ns)] + 0.3 * (rs.randn(ns) + 1j * rs.randn(ns))
        for nv in (0.37, 1e-310, 0.0):
            d_y = DeviceBuf(y.nbytes)
            _lib.check(lib.cpx_memcpy_h2d(d_y.ptr, _lib.ptr(y), y.nbytes))
            outs = []
            for scale in (None, -1.0, 1.0):
                d_l = DeviceBuf(ns * nb * 8)
                if scale is None:
                    _lib.check(lib.cpx_demod_soft_dev(md._device_handle(), d_y.ptr, ns, nv, d_l.ptr, None))
                else:
                    _lib.check(lib.cpx_demod_soft_scaled_dev(md._device_handle(), d_y.ptr, ns, nv, scale, d_l.ptr, None))
                o = np.empty(ns * nb)
                _lib.check(lib.cpx_memcpy_d2h(_lib.ptr(o), d_l.ptr, o.nbytes))
                outs.append(o)
            with np.errstate(invalid="ignore"):
                assert np.array_equal(outs[1], -outs[0], equal_nan=True)
                assert np.array_equal(outs[2], outs[0], equal_nan=True)
            if nv == 0.37:
                ref = oracle.demodulate(md.constellation, y, "soft", nv)
                assert np.max(np.abs(outs[0] - ref)) < 1e-9


def test_trace_mode_runs(gpu):
    """CPX_TRACE=1: the roctx ranges around the entry points are resolved at run time (dlopen) and must not change results."""
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from helpers import make_trellis; from commpy_amd.channelcoding import viterbi_decode;"
            "tr = make_trellis('t57'); x = np.random.RandomState(1).randint(0, 2, (5, 40)).astype(float);"
            "print(viterbi_decode(x, tr, None, 'hard').sum())") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                   os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for trace in ("0", "1"):
        env = dict(os.environ, CPX_TRACE=trace)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


def test_concurrent_host_threads_share_the_arena_safely(gpu):
    """ctypes releases the GIL, so two Python threads really are inside the library at once.  Both decode on the library's
    default stream with batch sizes that keep growing the shared scratch arena (LDPC staging / Viterbi two-kernel workspace /
    turbo slab): every result must equal the one computed alone (per-device issue lock, csrc/runtime.hip)."""
    import threading
    from commpy_amd import _lib
    from commpy_amd.channelcoding import RandInterlv, ldpc_bp_decode, turbo_decode, viterbi_decode
    rs = np.random.RandomState(12)
    p = ldpc_params("wimax1440")
    tr = make_trellis("k7_133_171")
    tr4 = make_trellis("rsc_legacy_4")
    il = RandInterlv(64, 7)
    sizes = [3, 40, 9, 130, 17, 260, 5, 400]
    ldpc_in = [rs.randn(b * 1440) * 2 + 1.5 for b in sizes]
    vit_in = [rs.randn(b * 2, 2 * 70) * 2 for b in sizes]
    tur_in = [[rs.randn(b, 64) for _ in range(3)] for b in sizes]
    _lib.viterbi_set_path("cw2")                                    # two-kernel form: takes arena slots 0 and 1 as well
    try:
        want_l = [ldpc_bp_decode(x.copy(), p, "MSA", 8) for x in ldpc_in]
        want_v = [viterbi_decode(x, tr, 20, "soft") for x in vit_in]
        want_t = [turbo_decode(s, a, b, tr4, 0.8, 3, il) for s, a, b in tur_in]
        errs = []

        def work(kind):
            try:
                for rep in range(6):
                    for i in range(len(sizes)):
                        if kind == 0:
                            d, o = ldpc_bp_decode(ldpc_in[i].copy(), p, "MSA", 8)
                            ok = np.array_equal(d, want_l[i][0]) and np.array_equal(o, want_l[i][1])
                        elif kind == 1:
                            ok = np.array_equal(viterbi_decode(vit_in[i], tr, 20, "soft"), want_v[i])
                        else:
                            s, a, b = tur_in[i]
                            ok = np.array_equal(turbo_decode(s, a, b, tr4, 0.8, 3, il), want_t[i])
                        if not ok:
                            errs.append((kind, rep, i))
            except Exception as e:                                  # noqa: BLE001 -- reported by the main thread
                errs.append((kind, repr(e)))

        th = [threading.Thread(target=work, args=(k,)) for k in (0, 1, 2, 1, 0)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs[:5]
    finally:
        _lib.viterbi_set_path(None)


@pytest.mark.parametrize("m,N0", [(64, 0.1), (64, 0.02), (256, 0.05), (256, 0.02), (16, 0.01)])  # the last: saturates below 600 too
def test_soft_demod_near_underflow_follows_the_reference(gpu, m, N0):
    """26 - 35 dB: LLRs of 600 - 745 and +-inf, where sums of e^{-d^2/N0} are down among the denormals.  The factorised
    (per-axis) sums round differently there, so such symbols are redone point by point in the reference's order
    (modulation.py:125-137): finiteness pattern equal to the oracle's, values to 1e-9 (found by scripts/fuzz_gpu.py)."""
    from commpy_amd.modulation import QAMModem
    md = QAMModem(m)
    rs = np.random.RandomState(m)
    ns = 6000
    y = md.constellation[rs.randint(0, md.m, ns)] + np.sqrt(N0 / 2) * (rs.randn(ns) + 1j * rs.randn(ns))
    soft = md.demodulate(y, "soft", N0)
    want = oracle.demodulate(md.constellation, y, "soft", N0)
    assert np.max(np.abs(want[np.isfinite(want)])) > (600 if m >= 64 else 400)
    assert np.array_equal(np.isfinite(soft), np.isfinite(want))
    assert np.array_equal(np.isnan(soft), np.isnan(want))
    inf = np.isinf(want)
    assert np.array_equal(soft[inf], want[inf])
    fin = np.isfinite(want)
    assert np.max(np.abs(soft[fin] - want[fin])) < 1e-9


def test_stream_destroy_retires_its_scratch_blocks(gpu):
    """A caller's stream grows scratch blocks of its own (the arena is keyed by device and stream).  cpx_stream_destroy frees them:
    round 5 found cpx_release_workspace() synchronising a stream that DeviceGroup.close() had destroyed -- an abort from inside the
    HIP runtime, visible only when the collectives tests ran before a test that releases the workspace."""
    from commpy_amd import _lib
    from commpy_amd.channelcoding.ldpc import _device_code
    from commpy_amd.devicelink import DeviceBuf
    lib = _lib.load()
    p = ldpc_params("wimax1440")
    code = _device_code(p)
    B, n = 300, 1440
    llr = np.random.RandomState(3).randn(B, n) * 2 + 1.5
    outs = []
    for use_stream in (True, False, True):
        s = ctypes.c_void_p()
        if use_stream:
            _lib.check(lib.cpx_stream_create(ctypes.byref(s)))
        d_llr = DeviceBuf.from_array(llr)
        d_dec, d_out, d_it = DeviceBuf(B * n), DeviceBuf(B * n * 8), DeviceBuf(B * 4)
        for alg in (1, 0):
            _lib.check(lib.cpx_ldpc_bp_decode_batch_bm_dev(code, d_llr.ptr, B, alg, 12, d_dec.ptr, d_out.ptr, d_it.ptr, s if use_stream else None))
        _lib.check(lib.cpx_stream_sync(s if use_stream else None))
        outs.append((d_dec.to_array((B, n), np.int8), d_out.to_array((B, n), np.float64), d_it.to_array((B,), np.int32)))
        if use_stream:
            _lib.check(lib.cpx_stream_destroy(s))
        _lib.check(lib.cpx_release_workspace())                       # aborted here with the destroyed stream's entries left behind
        for b in (d_llr, d_dec, d_out, d_it):
            b.free()
    for o in outs[1:]:
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(o, outs[0]))


def test_sclk_probes_own_their_result_slots(gpu):
    """Round 6 (advisor): every outstanding shader-clock probe owns one of 16 result slots of its device -- two probes started before a
    read no longer alias the same four words --, the 17th start is refused (CPX_ELIMIT -> ValueError), an unread probe goes back with
    cpx_sclk_probe_destroy, and overlapping probes report plausible, separate intervals."""
    from commpy_amd import _lib
    lib = _lib.load()
    probes = []
    for i in range(16):
        p = ctypes.c_void_p()
        _lib.check(lib.cpx_sclk_probe_start(ctypes.byref(p), 0.5 + 0.25 * i))
        probes.append(p)
    extra = ctypes.c_void_p()
    with pytest.raises(ValueError):
        _lib.check(lib.cpx_sclk_probe_start(ctypes.byref(extra), 0.5))
    got = []
    for p in probes[:8]:
        mhz, ival = ctypes.c_double(), ctypes.c_double()
        _lib.check(lib.cpx_sclk_probe_read(p, ctypes.byref(mhz), ctypes.byref(ival)))
        got.append((mhz.value, ival.value))
    for p in probes[8:]:
        _lib.check(lib.cpx_sclk_probe_destroy(p))
    for i, (mhz, ival) in enumerate(got):                          # each probe measured ITS interval (they ran one after the other on the probe stream)
        assert 100.0 < mhz < 4000.0, got
        assert 0.5 + 0.25 * i <= ival < 0.5 + 0.25 * i + 0.2, got
    p = ctypes.c_void_p()                                           # all slots free again
    _lib.check(lib.cpx_sclk_probe_start(ctypes.byref(p), 0.2))
    _lib.check(lib.cpx_sclk_probe_read(p, None, None))
