"""Host-side logic of commpy_amd (no GPU): trellis tables, encoders, bit helpers, interleaver,
constellations, LDPC design files -- against the reference's own golden tables
(commpy/channelcoding/tests/test_convcode.py:23-111, commpy/tests/test_utilities.py:12-13,
commpy/tests/test_modulation.py:164-174) and the fixtures generated from the live reference."""
import os
import warnings

import numpy as np
import pytest

from helpers import TRELLIS_SPECS, golden, ldpc_params, make_trellis

from commpy_amd.channelcoding.convcode import Trellis, conv_encode, depuncturing, puncturing
from commpy_amd.channelcoding.interleavers import RandInterlv
from commpy_amd.utilities import bitarray2dec, dec2bitarray, euclid_dist, hamming_dist

# literal tables restated from the reference's test file (test_convcode.py:27-111)
REF_TABLES = {
    "t57": ([[0, 2], [0, 2], [1, 3], [1, 3]], [[0, 3], [3, 0], [1, 2], [2, 1]], [0, 0, 0, 0, 1, 1, 0, 1]),
    "rsc_legacy_4": ([[0, 2], [2, 0], [1, 3], [3, 1]], [[0, 3], [0, 3], [1, 2], [1, 2]], [0, 0, 0, 0, 1, 1, 0, 1]),
    "k2_default": ([[0, 1, 4, 5]] * 4 + [[2, 3, 6, 7]] * 4,
                   [[0, 1, 6, 7], [3, 2, 5, 4], [6, 7, 0, 1], [5, 4, 3, 2], [2, 3, 4, 5], [1, 0, 7, 6], [4, 5, 2, 3],
                    [7, 6, 1, 0]], [0, 0, 0, 1, 1, 0]),
    "k2_lsb": ([[0, 1, 4, 5]] * 4 + [[2, 3, 6, 7]] * 4,
               [[0, 1, 6, 7], [3, 2, 5, 4], [6, 7, 0, 1], [5, 4, 3, 2], [2, 3, 4, 5], [1, 0, 7, 6], [4, 5, 2, 3],
                [7, 6, 1, 0]], [0, 0, 0, 1, 1, 0]),
    "k2_rsc_matrix": ([[0, 1, 1, 0], [2, 3, 3, 2], [3, 2, 2, 3], [1, 0, 0, 1]],
                      [[0, 3, 4, 7], [1, 2, 5, 6], [0, 3, 4, 7], [1, 2, 5, 6]], [0, 0, 0, 1, 0, 0]),
}


@pytest.mark.parametrize("name", list(REF_TABLES))
def test_trellis_reference_golden_tables(name):
    nxt, out, enc = REF_TABLES[name]
    tr = make_trellis(name)
    assert np.array_equal(tr.next_state_table, nxt)
    assert np.array_equal(tr.output_table, out)
    assert np.array_equal(conv_encode(np.array((0, 0, 1, 0)), tr, "cont"), enc)


@pytest.mark.parametrize("spec", TRELLIS_SPECS, ids=[s[0] for s in TRELLIS_SPECS])
def test_trellis_and_encoder_vs_live_reference(spec):
    g, e = golden("trellis"), golden("conv_encode")
    name = spec[0]
    tr = make_trellis(name)
    assert np.array_equal(tr.next_state_table, g[name + "__next"])
    assert np.array_equal(tr.output_table, g[name + "__out"])
    assert [tr.k, tr.n, tr.total_memory, tr.number_states, tr.number_inputs] == list(g[name + "__kn"])
    assert np.array_equal(conv_encode(e[name + "__msg"], tr), e[name + "__term"])
    assert np.array_equal(conv_encode(e[name + "__msg"], tr, "cont"), e[name + "__cont"])


def test_trellis_quirks():
    # B1: Wifi80211's decimal generators (133, 171) silently become (5, 43)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = Trellis(np.array([6]), np.array([[133, 171]]))
        b = Trellis(np.array([6]), np.array([[5, 43]]))
    assert np.array_equal(a.output_table, b.output_table) and np.array_equal(a.next_state_table, b.next_state_table)
    # legacy int feedback warns and mutates the caller's g_matrix for rsc codes
    g = np.array([[1, 7]])
    with pytest.warns(DeprecationWarning):
        Trellis(np.array([2]), g, 5, "rsc")
    assert g[0][0] == 5
    with pytest.raises(ValueError):
        Trellis(np.array([2]), np.array([[5, 7]]), polynomial_format="XSB")
    # B2: legacy and matrix constructions give different RSC output tables
    assert not np.array_equal(make_trellis("rsc_legacy_4").output_table, make_trellis("rsc_matrix_4").output_table)


def test_bit_helpers():
    g = golden("trellis")
    assert np.array_equal(dec2bitarray(17, 8), [0, 0, 0, 1, 0, 0, 0, 1])           # test_utilities.py:12
    assert np.array_equal(dec2bitarray((17, 12), 5), [1, 0, 0, 0, 1, 0, 1, 1, 0, 0])  # test_utilities.py:13
    assert np.array_equal(dec2bitarray(133, 7), g["dec2bit_133_7"])                # wrap quirk
    assert np.array_equal(dec2bitarray(171, 7), g["dec2bit_171_7"])
    assert dec2bitarray(5, 4).dtype == np.int8
    assert bitarray2dec(dec2bitarray(45, 9)) == 45
    assert hamming_dist(np.array([1, 0, 1]), np.array([0, 0, 1])) == 1
    assert euclid_dist(np.array([1., 2.]), np.array([0., 0.])) == 5.0


def test_puncturing():
    e = golden("conv_encode")
    for nm in ("p23", "p34", "p56"):
        pu = puncturing(e["punct_" + nm + "__msg"], e["punct_" + nm + "__vec"])
        assert np.array_equal(pu, e["punct_" + nm + "__punctured"])
        de = depuncturing(pu.astype(float) * 2 - 1, e["punct_" + nm + "__vec"], 120)
        assert np.array_equal(de, e["punct_" + nm + "__depunctured"])


def test_interleaver_and_turbo_encode():
    from commpy_amd.channelcoding.turbo import turbo_encode
    g = golden("map_turbo")
    il = RandInterlv(16, 7)
    assert np.array_equal(il.p_array, g["randinterlv_16_7"])
    x = np.arange(16) * 1.5
    assert np.array_equal(il.deinterlv(il.interlv(x)), x)
    for nm in g["turbo_names"]:
        key, tname, N, nv, iters = str(nm).split("|")
        tr = make_trellis(tname)
        il = RandInterlv(int(N), 1234)
        assert np.array_equal(il.p_array, g[key + "__perm"])
        s, p1, p2 = turbo_encode(g[key + "__msg"], tr, tr, il)
        assert np.array_equal(s, g[key + "__enc_s"]) and np.array_equal(p1, g[key + "__enc_p1"])
        assert np.array_equal(p2[:int(N)], g[key + "__enc_p2"])


def test_modem_constellations():
    from commpy_amd.modulation import Modem, PSKModem, QAMModem
    g = golden("demod")
    mods = {"qam4": QAMModem(4), "qam16": QAMModem(16), "qam64": QAMModem(64), "qam256": QAMModem(256),
            "psk2": PSKModem(2), "psk4": PSKModem(4), "psk8": PSKModem(8), "psk16": PSKModem(16)}
    for k, m in mods.items():
        assert np.array_equal(m.constellation, g[k + "__const"]), k
        assert m.Es == g[k + "__Es"], k
        assert np.array_equal(m.modulate(g[k + "__bits"]), g[k + "__sym"]), k
    # Es values pinned by the reference's own test (test_modulation.py:164-174): QAM Es = 2(m-1)/3, PSK Es = 1
    assert np.isclose(QAMModem(64).Es, 42) and np.isclose(QAMModem(4).Es, 2) and np.isclose(PSKModem(8).Es, 1)
    assert list(QAMModem(64).constellation[:4]) == [-7 - 7j, -7 - 5j, -7 - 1j, -7 - 3j]
    with pytest.raises(ValueError):
        QAMModem(32)
    with pytest.raises(ValueError):
        PSKModem(6)
    with pytest.raises(ValueError):
        Modem([1, 2, 3])


def test_ldpc_design_files(tmp_path):
    from commpy_amd.channelcoding.ldpc import (build_matrix, get_ldpc_code_params, triang_ldpc_systematic_encode,
                                               write_ldpc_params)
    here = os.path.dirname(os.path.abspath(__file__))
    own = os.path.join(here, "..", "commpy_amd", "channelcoding", "designs", "ldpc", "ieee80211n", "1944.1296.txt")
    p = get_ldpc_code_params(own)
    q = ldpc_params("n1944")
    for k in q:
        assert np.array_equal(p[k], q[k]), k
    assert p["cnode_adj_list"].dtype == np.int32
    # write -> read round trip (test_ldpc.py:68-75)
    rs = np.random.RandomState(4)
    H = rs.choice((0, 1), (30, 60))
    H[:, 0] = 1
    H[0, :] = 1
    path = str(tmp_path / "matrix.txt")
    write_ldpc_params(H, path)
    back = get_ldpc_code_params(path, True)
    assert np.array_equal(back["parity_check_matrix"].toarray(), H)
    # systematic encoder on a WiMax code: codeword of the reference, zero syndrome (test_ldpc.py:77-92)
    g = golden("ldpc")
    w = ldpc_params("wimax1440")
    coded = triang_ldpc_systematic_encode(g["enc1440__msg"], w)
    assert np.array_equal(coded, g["enc1440__coded"])
    assert not (w["parity_check_matrix"].dot(coded) % 2).any()
    with pytest.raises(ValueError):
        triang_ldpc_systematic_encode(np.array([0, 1]), w, False)


@pytest.mark.timeout(240)
def test_bench_cpu_baseline_legs():
    """bench.py's cpu_baseline: the 'reference' leg (the unmodified CommPy decoder on a process pool) where the reference is present
    -- the build container --, the C-port leg and the NumPy secondary everywhere; all on four codewords of the unquantised config-2
    fixture, whose stored decodes are the live reference's."""
    import bench                                                   # repo root is on sys.path (conftest); the pool's spawned workers import it by name
    from helpers import golden
    from test_oracle_golden import TableTrellis
    g = golden("viterbi_c2u")
    llr = np.ascontiguousarray(g["llr"][:4])
    dec = np.unpackbits(g["dec"], axis=1)[:4, :1030].astype(np.int64)
    res = bench.cpu_baseline(TableTrellis("k7_133_171"), llr, dec, budget_s=0.3)
    port = res if res["kind"] == "port" else res["port"]
    assert port["kind"] == "port" and port["value"] > 1e4
    assert res["secondary"]["sample"].endswith("True")
    if bench._find_reference():
        assert res["kind"] == "reference" and res["cores"] == 4
        assert "bits differing from the engine's on these codewords: 0" in res["sample"]
        assert 100 < res["per_core"] < 1e5


@pytest.mark.timeout(240)
def test_bench_times_the_reference_from_oracle_ref(monkeypatch):
    """The copy of the reference's six hot-path files that travels to the GPU box (oracle/_ref, placed by build() through
    oracle/make_ref.py) is complete enough to run the reference's own viterbi_decode in a worker process, is byte-identical
    to the checkout where both exist, and gives the stored live-reference decodes."""
    import bench
    from oracle import make_ref
    from helpers import golden
    from test_oracle_golden import TableTrellis
    ref = make_ref.make()
    if ref is None:
        pytest.skip("oracle/_ref not populated (build() runs make_ref where /root/reference exists)")
    if os.path.isdir(make_ref.DEFAULT_SRC):
        assert make_ref.check()
    assert ref.endswith(".zip")                                    # one archive, a sys.path entry (zipimport)
    prov = open(os.path.join(os.path.dirname(ref), "PROVENANCE.txt")).read()
    assert all(f in prov for f in make_ref.FILES) and os.path.isfile(os.path.join(os.path.dirname(ref), "LICENSE.txt"))
    monkeypatch.setenv("CPX_REFERENCE_PATH", ref)
    assert bench._find_reference() == ref
    g = golden("viterbi_c2u")
    llr = np.ascontiguousarray(g["llr"][:2])
    dec = np.unpackbits(g["dec"], axis=1)[:2, :1030].astype(np.int64)
    res = bench.cpu_baseline(TableTrellis("k7_133_171"), llr, dec, budget_s=0.2)
    assert res["kind"] == "reference" and res["cores"] == 2 and ref in res["sample"]
    assert "bits differing from the engine's on these codewords: 0" in res["sample"]


# ---- round 4: trellises beyond the reference's own constructor, ready-made tables ------------------------------------------------
def test_trellis_large_memory_and_many_outputs_match_a_shift_register_model():
    """K = 9 / K = 10 and rate-1/8 tables (the reference's Trellis overflows an int8 there under NumPy 2): state = last `mem`
    inputs, most recent first; 'MSB' polynomial format: bit w of g taps D^w, i.e. bit mem - w of the register word."""
    from commpy_amd.channelcoding import Trellis
    for mem, gens in ((8, (0o561, 0o753)), (9, (0o1167, 0o1545)), (3, (0o17, 0o15, 0o13, 0o11, 0o7, 0o5, 0o16, 0o12))):
        t = Trellis(np.array([mem]), np.array([list(gens)]))
        assert (t.number_states, t.number_inputs, t.k, t.n) == (1 << mem, 2, 1, len(gens))
        rev = [int(format(int(g), "0%db" % (mem + 1))[::-1], 2) for g in gens]
        for st in range(0, 1 << mem, 7):
            for u in (0, 1):
                reg = (u << mem) | st
                o = 0
                for gr in rev:
                    o = 2 * o + (bin(reg & gr).count("1") & 1)
                assert t.output_table[st, u] == o and t.next_state_table[st, u] == reg >> 1


def test_trellis_from_tables():
    from commpy_amd.channelcoding import Trellis
    t = Trellis(np.array([2]), np.array([[5, 7]]))
    u = Trellis.from_tables(t.k, t.n, t.total_memory, t.next_state_table.tolist(), t.output_table)
    assert (u.k, u.n, u.total_memory, u.number_states, u.number_inputs) == (1, 2, 2, 4, 2)
    assert np.array_equal(u.next_state_table, t.next_state_table) and np.array_equal(u.output_table, t.output_table)
    import pytest
    with pytest.raises(ValueError):
        Trellis.from_tables(2, 2, 2, t.next_state_table, t.output_table)      # 2 ** k columns expected


def test_demod_tables_in_the_kernel_source_are_what_their_comments_say():
    """csrc/demod.hip carries three 32-entry tables as hex literals (2^(j/32); 1/c_i for the centres of 32 mantissa intervals; -log of the
    ROUNDED 1/c_i): recomputed here in extended precision -- a pasted constant that is off by one digit would still pass the GPU parity
    tests at 1e-5 and sit in the kernel unnoticed."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "commpy_amd", "csrc", "demod.hip")).read()
    body = src[src.index("__device__ const double DEMOD_TAB[96] = {"):]
    body = body[:body.index("};")]
    vals = [float.fromhex(v) for v in re.findall(r"-?0x1\.[0-9a-f]+p[+-]\d+", body)]
    assert len(vals) == 96
    ld = np.longdouble
    for j in range(32):
        assert vals[j] == float(ld(2) ** (ld(j) / ld(32))), j
        c = ld(0.5) + (ld(j) + ld(0.5)) / ld(64)
        assert vals[32 + j] == float(ld(1) / c), j
        want = float(-np.log(ld(vals[32 + j])))
        assert abs(vals[64 + j] - want) <= abs(want) * 2.3e-16, j
    consts = dict(re.findall(r"(INV_L32|L32_HI|L32_LO|LN2) = (0x1\.[0-9a-f]+p[+-]\d+)", src))
    l32 = np.log(ld(2)) / ld(32)
    assert float.fromhex(consts["L32_HI"]) == float(l32)
    assert abs(float.fromhex(consts["L32_LO"]) - float(l32 - ld(float(l32)))) < 1e-21
    assert float.fromhex(consts["INV_L32"]) == float(ld(32) / np.log(ld(2)))
    assert float.fromhex(consts["LN2"]) == float(np.log(ld(2)))


def test_other_configs_quotes_counters_only_from_the_same_build(tmp_path, monkeypatch):
    """benchmarks/other_configs.py::pmc_fields (round 6): `traffic` = sum over a step's kernels of launches x bytes per launch, `valu` of the
    kernel a step spends most of its time in -- and nothing at all when the PMC file was recorded by another build, when a kernel of the
    step is missing from it, or when there is no file."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from benchmarks import other_configs as oc
    monkeypatch.setattr(oc, "PMC_DIR", str(tmp_path))
    monkeypatch.setattr(oc._lib, "build_id", lambda: {"full": "abc", "viterbi": "v"})
    none = {"traffic": None, "traffic_source": None, "valu": None}
    assert oc.pmc_fields("turbo", {"k_pass": 12}) == none                                  # no file
    doc = {"build_id": {"full": "abc"}, "git_head": "deadbeefcafe", "fetch_scale": 2.0,
           "kernels": {"k_pass<2, true>": {"traffic_bytes_per_launch": 100.0, "duration_ns_avg": 10.0, "calls": 12,
                                           "valu": {"busy_frac": 0.5}, "SQ_LDS_BANK_CONFLICT": 2.0, "SQ_ACTIVE_INST_LDS": 8.0},
                       "k_init": {"traffic_bytes_per_launch": 1000.0, "duration_ns_avg": 50.0, "calls": 1, "valu": {"busy_frac": 0.9}}}}
    path = tmp_path / "r06_oc_turbo_pmc.json"
    path.write_text(json.dumps(doc))
    got = oc.pmc_fields("turbo", {"k_pass": 12, "k_init": 1})
    assert got["traffic"] == 12 * 100.0 + 1000.0
    assert got["valu"] == {"busy_frac": 0.5}                                               # 12 x 10 ns > 1 x 50 ns: the pass dominates
    assert got["per_kernel"]["k_pass<2, true>"]["lds_bank_conflict_frac_of_lds_active"] == 0.25
    assert "r06_oc_turbo_pmc.json" in got["traffic_source"] and "abc" in got["traffic_source"]
    assert oc.pmc_fields("turbo", {"k_pass": 12, "k_missing": 1}) == none                  # a kernel of the step is not in the file
    doc["build_id"]["full"] = "other"
    path.write_text(json.dumps(doc))
    assert oc.pmc_fields("turbo", {"k_pass": 12, "k_init": 1}) == none                     # recorded by another build
