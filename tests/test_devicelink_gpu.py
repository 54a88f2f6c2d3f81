"""Device-side link stages ("next" rows): encoder / modulator bit-exact against the host mirror and the
reference goldens, Philox noise statistics, and the GPU-resident Wifi80211 chain against reference BER points."""
import ctypes

import numpy as np
import pytest

from helpers import TRELLIS_SPECS, golden, make_trellis

pytestmark = pytest.mark.gpu


def test_conv_encode_gpu_bit_exact(gpu):
    from commpy_amd.channelcoding import conv_encode, conv_encode_batch
    from commpy_amd.devicelink import conv_encode_gpu
    e = golden("conv_encode")
    rs = np.random.RandomState(3)
    for spec in TRELLIS_SPECS:
        name = spec[0]
        tr = make_trellis(name)
        # the reference's own vectors
        assert np.array_equal(conv_encode_gpu(e[name + "__msg"], tr)[0], e[name + "__term"]), name
        assert np.array_equal(conv_encode_gpu(e[name + "__msg"], tr, "cont")[0], e[name + "__cont"]), name
        msgs = rs.randint(0, 2, (300, 48))
        for term in ("term", "cont"):
            assert np.array_equal(conv_encode_gpu(msgs, tr, term), conv_encode_batch(msgs, tr, term)), (name, term)
    assert np.array_equal(conv_encode_gpu(np.array([0, 0, 1, 0]), make_trellis("t57"), "cont")[0],
                          [0, 0, 0, 0, 1, 1, 0, 1])            # test_convcode.py:36


def test_modulate_gpu_exact(gpu):
    from commpy_amd.devicelink import modulate_gpu
    from commpy_amd.modulation import PSKModem, QAMModem
    rs = np.random.RandomState(4)
    for md in (QAMModem(4), QAMModem(16), QAMModem(64), QAMModem(256), PSKModem(2), PSKModem(8)):
        bits = rs.randint(0, 2, 1000 * md.num_bits_symbol)
        assert np.array_equal(modulate_gpu(md, bits), md.modulate(bits))


def test_philox_bits_and_awgn_statistics(gpu):
    from commpy_amd import _lib
    from commpy_amd.devicelink import DeviceBuf
    lib = _lib.load()
    n = 1 << 20
    d_bits = DeviceBuf(n)
    _lib.check(lib.cpx_random_bits_dev(d_bits.ptr, n, 7, 1, None))
    _lib.check(lib.cpx_stream_sync(None))
    bits = d_bits.to_array((n,), np.uint8)
    assert set(np.unique(bits)) == {0, 1} and abs(bits.mean() - 0.5) < 3e-3
    assert abs(np.mean(bits[:-1] ^ bits[1:]) - 0.5) < 3e-3                     # no obvious serial correlation
    x = np.zeros(n, complex)
    d_x, d_y = DeviceBuf.from_array(x), DeviceBuf(n * 16)
    _lib.check(lib.cpx_awgn_dev(d_x.ptr, n, 2.0, 0.5, 7, 2, d_y.ptr, None))
    _lib.check(lib.cpx_stream_sync(None))
    y = d_y.to_array((n,), np.complex128)
    assert abs(y.real.mean()) < 0.01 and abs(y.imag.mean()) < 0.01
    assert abs(y.real.std() - 2.0) < 0.01 and abs(y.imag.std() - 0.5) < 0.005
    assert abs(np.corrcoef(y.real, y.imag)[0, 1]) < 5e-3
    assert abs(np.mean(y.real ** 4) / 16.0 - 3.0) < 0.05                        # Gaussian kurtosis
    _lib.check(lib.cpx_awgn_dev(d_x.ptr, n, 2.0, 0.5, 7, 3, d_y.ptr, None))     # another stream id -> another draw
    _lib.check(lib.cpx_stream_sync(None))
    assert not np.array_equal(y, d_y.to_array((n,), np.complex128))


@pytest.mark.parametrize("gname,mcs", [("octal", 1), ("octal", 5), ("octal", 3), ("decimal", 1)])
def test_device_wifi_link_overlays_reference(gpu, gname, mcs):
    from commpy_amd.devicelink import DeviceWifiLink
    g = golden("wifi")
    key = "w_%s_mcs%d" % (gname, mcs)
    snrs, ref = g[key + "__snrs"], g[key + "__ber"]
    link = DeviceWifiLink(mcs, 600, generator_matrix=[[0o133, 0o171]] if gname == "octal" else None, seed=11 + mcs)
    ref_bits = int(g[key + "__tx"]) * 600
    # per-point calls, and the whole sweep through one Viterbi call (the large-batch kernel from 29 492 frames)
    for bers in (link.ber_sweep(snrs, 600 * 2048), link.ber_sweep_batched(snrs, 600 * 8192)):
        for s, b, r in zip(snrs, bers, ref):
            ref_errors = r * ref_bits
            if ref_errors >= 100:
                assert r / 2 <= b <= 2 * r, (key, s, b, r)
            elif ref_errors >= 10:
                assert r / 4 <= b <= 4 * r, (key, s, b, r)
            else:
                assert b <= max(8 * r, 100.0 / ref_bits), (key, s, b, r)


def test_device_wifi_link_matches_host_pipeline(gpu):
    """Same chain through the host-orchestrated Wifi80211 (NumPy RNG) and the device-resident link (Philox)."""
    from commpy_amd.channels import SISOFlatChannel
    from commpy_amd.devicelink import DeviceWifiLink
    from commpy_amd.wifi80211 import Wifi80211
    np.random.seed(5)
    snrs = np.array([5.0, 6.0])
    host = Wifi80211(1, generator_matrix=[[0o133, 0o171]]).link_performance(
        SISOFlatChannel(fading_param=(1 + 0j, 0j)), snrs, 1500, 1, 600, stop_on_surpass_error=False)[0]
    dev = DeviceWifiLink(1, 600, generator_matrix=[[0o133, 0o171]], seed=3).ber_sweep(snrs, 600 * 6000)
    assert np.all(dev > 0) and np.all(host > 0)
    assert np.all(np.abs(np.log(dev / host)) < np.log(1.6)), (dev, host)


def test_device_wifi_link_batched_sweep_noiseless_and_equal_paths(gpu):
    """The batched sweep decodes every frame of every point (very high SNR -> zero errors), and its BER does not depend
    on which Viterbi kernels decode it (same random streams, cpx_viterbi_set_path('wave') vs automatic)."""
    from commpy_amd.devicelink import DeviceWifiLink
    snrs = np.array([40.0, 41.0, 42.0])
    assert not DeviceWifiLink(5, 1200, generator_matrix=[[0o133, 0o171]], seed=9).ber_sweep_batched(snrs, 1200 * 20000).any()
    snrs = np.array([14.0, 15.0, 16.0])
    res = {}
    from commpy_amd import _lib
    for path in ("wave", "auto"):
        _lib.viterbi_set_path(path)
        try:
            res[path] = DeviceWifiLink(5, 1200, generator_matrix=[[0o133, 0o171]], seed=9).ber_sweep_batched(snrs, 1200 * 20000)
            assert _lib.viterbi_last_path() == ("wave" if path == "wave" else "fused")    # 60000 frames: one round
        finally:
            _lib.viterbi_set_path(None)
    assert np.array_equal(res["wave"], res["auto"]) and res["auto"][0] > 0


def test_device_wifi_link_noiseless(gpu):
    from commpy_amd.devicelink import DeviceWifiLink
    for mcs in (0, 2, 4, 5, 7, 9):
        link = DeviceWifiLink(mcs, 1200, frame_aggregation=2, generator_matrix=[[0o133, 0o171]])
        errs = link.run_batch(70.0, 33)
        assert errs.shape == (33, 2) and not errs.any(), mcs
