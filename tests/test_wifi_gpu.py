"""BASELINE config 5 semantics on the GPU decoders: Wifi80211 link BER against points measured with the
live reference (tests/golden/wifi.npz): DETERMINISTICALLY, error count by error count, on the one-transmission-at-a-time path
(round 6), and as a statistical overlay on the batched path, for the shipped decimal generators (quirk B1: catastrophic (5,43)
code) and for the intended octal (133,171) ones."""
import numpy as np
import pytest

from helpers import golden

pytestmark = pytest.mark.gpu


RUNS = [("octal", 1), ("octal", 5), ("octal", 3), ("decimal", 1), ("decimal", 5), ("decimal", 3)]


@pytest.mark.parametrize("gname,mcs", RUNS)
def test_wifi80211_error_counts_equal_the_reference(gpu, gname, mcs):
    """Config 5, deterministic: the whole receive chain -- soft demodulator -> depuncturing -> soft Viterbi -- against the live
    reference, transmission by transmission.  tests/golden/wifi.npz holds the reference's per-transmission error counts of
    ``Wifi80211(mcs).link_performance(channel, snrs, tx, 1, 600, stop_on_surpass_error=False)`` under ``np.random.seed(2024 + mcs)``
    (make_golden.gen_wifi; wifi80211.py:132-216, links.py:155-267).  With ``tx_batch = 1`` this package draws NumPy's global stream
    exactly like the reference (message, noise real part, imaginary part, the two fading draws; commpy_amd/links.py::_run_block), so
    every count has to be EQUAL: 64 transmissions x 2-3 SNR points per case, shipped decimal generators (quirk B1) and octal ones."""
    from commpy_amd.channels import SISOFlatChannel
    from commpy_amd.wifi80211 import Wifi80211
    g = golden("wifi")
    key = "w_%s_mcs%d" % (gname, mcs)
    snrs, tx = g[key + "__snrs"], int(g[key + "__tx"])
    w = Wifi80211(mcs, generator_matrix=[[0o133, 0o171]] if gname == "octal" else None)
    w.tx_batch = 1
    ch = SISOFlatChannel(fading_param=(1 + 0j, 0j))
    np.random.seed(2024 + mcs)
    bers, bes, ces, ncs = w.link_performance(ch, snrs, tx, 1, 600, stop_on_surpass_error=False)
    assert np.array_equal(bes, g[key + "__bes"]), (key, np.argwhere(bes != g[key + "__bes"])[:5])
    assert np.array_equal(ces, g[key + "__ces"]) and np.array_equal(ncs, g[key + "__ncs"])
    assert np.array_equal(bers, g[key + "__ber"])


@pytest.mark.parametrize("gname,mcs", RUNS[:5])
def test_wifi80211_ber_overlays_reference(gpu, gname, mcs):
    """The BATCHED path (another random stream, 256 transmissions per point) against the same reference points, statistically: a
    two-sample test on the per-transmission error counts, whose spread is measured on both sides (error events of a Viterbi decoder
    are bursts; round 5 used factor-2 ... 8 bands around reference points of 12 - 30 transmissions)."""
    from commpy_amd.channels import SISOFlatChannel
    from commpy_amd.wifi80211 import Wifi80211
    g = golden("wifi")
    key = "w_%s_mcs%d" % (gname, mcs)
    snrs, ref_bes = g[key + "__snrs"], g[key + "__bes"]
    np.random.seed(99 + mcs)
    w = Wifi80211(mcs, generator_matrix=[[0o133, 0o171]] if gname == "octal" else None)
    ch = SISOFlatChannel(fading_param=(1 + 0j, 0j))
    tx = 256
    bers, bes, ces, ncs = w.link_performance(ch, snrs, tx, 1, 600, stop_on_surpass_error=False)
    for i, s in enumerate(snrs):
        if i > 0 and g[key + "__ber"][i - 1] == 0:           # the reference's sweep had stopped (links.py:262)
            break
        a, b = bes[i].astype(float), ref_bes[i].astype(float)
        se = np.sqrt(a.var(ddof=1) / a.size + b.var(ddof=1) / b.size)
        # 4.5 standard errors + half an error per transmission for the points where both samples are (almost) error free
        assert abs(a.mean() - b.mean()) <= 4.5 * se + 0.5, (key, float(s), a.mean(), b.mean(), se)
    # BER decreases with SNR unless the code is catastrophic and saturated
    if gname == "octal":
        assert bers[0] >= bers[-1]


def test_wifi80211_noiseless_is_error_free(gpu):
    from commpy_amd.channels import SISOFlatChannel
    from commpy_amd.wifi80211 import Wifi80211
    np.random.seed(1)
    for mcs in (0, 2, 4, 7, 9):
        w = Wifi80211(mcs, generator_matrix=[[0o133, 0o171]])
        ch = SISOFlatChannel(fading_param=(1 + 0j, 0j))
        bers, bes, ces, ncs = w.link_performance(ch, np.array([60.0]), 8, 1, 1200, frame_aggregation=2,
                                                 stop_on_surpass_error=False)
        assert bers[0] == 0 and ncs[0, 0] == 2, mcs
