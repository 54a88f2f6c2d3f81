#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the LIVE reference.

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py [--only NAME]

The script imports the unmodified reference (veeresht/CommPy 0.8.0) read-only
from /root/reference, feeds it seeded synthetic inputs and stores
inputs + reference outputs.  Nothing here is imported by the product;
tests/ load the .npz files to pin the oracle (oracle/) and the HIP path.

Reference entry points exercised (file:line in /root/reference):
  Trellis                 commpy/channelcoding/convcode.py:117
  conv_encode             commpy/channelcoding/convcode.py:475
  viterbi_decode          commpy/channelcoding/convcode.py:661
  puncturing/depuncturing commpy/channelcoding/convcode.py:752,777
  map_decode/turbo_decode commpy/channelcoding/turbo.py:163,254
  turbo_encode            commpy/channelcoding/turbo.py:14
  RandInterlv             commpy/channelcoding/interleavers.py:50
  get_ldpc_code_params    commpy/channelcoding/ldpc.py:51
  ldpc_bp_decode          commpy/channelcoding/ldpc.py:144
  Modem.demodulate        commpy/modulation.py:100
"""
import argparse
import os
import sys
import time
import warnings

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

warnings.simplefilter("ignore")

from commpy.channelcoding.convcode import (Trellis, conv_encode, viterbi_decode,  # noqa: E402
                                           puncturing, depuncturing)
from commpy.channelcoding.turbo import turbo_encode, map_decode, turbo_decode  # noqa: E402
from commpy.channelcoding.interleavers import RandInterlv  # noqa: E402
from commpy.channelcoding.ldpc import (get_ldpc_code_params, ldpc_bp_decode,  # noqa: E402
                                       write_ldpc_params, triang_ldpc_systematic_encode)
from commpy.modulation import PSKModem, QAMModem, Modem  # noqa: E402
from commpy.utilities import dec2bitarray, bitarray2dec  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %s (%.1f kB, %d arrays)" % (path, os.path.getsize(path) / 1e3, len(arrs)))


# ----------------------------------------------------------------------------------------------
# Trellis constructions: (name, memory, g_matrix, feedback, code_type, polynomial_format)
# The first five are the reference's own test trellises (test_convcode.py:23-111).
# ----------------------------------------------------------------------------------------------
def trellis_specs():
    return [
        ("t57", [2], [[5, 7]], None, "default", "MSB"),
        ("rsc_legacy_4", [2], [[1, 7]], 5, "rsc", "MSB"),
        ("k2_default", [2, 1], [[5, 7, 0], [0, 2, 3]], None, "default", "MSB"),
        ("k2_lsb", [2, 1], [[5, 7, 0], [0, 2, 6]], None, "default", "LSB"),
        ("k2_rsc_matrix", [1, 1], [[1, 0, 0], [0, 1, 3]], [[2, 2], [3, 1]], "rsc", "MSB"),
        ("k7_133_171", [6], [[0o133, 0o171]], None, "default", "MSB"),
        ("wifi_decimal_133_171", [6], [[133, 171]], None, "default", "MSB"),
        ("rsc_legacy_8", [3], [[1, 0o15]], 0o13, "rsc", "MSB"),
        ("rsc_matrix_4", [2], [[1, 7]], [[5]], "rsc", "MSB"),
        ("r13_k4", [3], [[0o13, 0o15, 0o17]], None, "default", "MSB"),
        ("k5_23_35", [4], [[0o23, 0o35]], None, "default", "MSB"),
        ("k8_247_371", [7], [[0o247, 0o371]], None, "default", "MSB"),
    ]


def make_trellis(spec):
    name, mem, g, fb, ctype, fmt = spec
    mem = np.array(mem)
    g = np.array(g)
    if fb is None:
        return Trellis(mem, g, code_type=ctype, polynomial_format=fmt)
    if isinstance(fb, int):
        return Trellis(mem, g, fb, ctype)
    return Trellis(mem, g, np.array(fb), ctype, polynomial_format=fmt)


def gen_trellis():
    out = {}
    for spec in trellis_specs():
        tr = make_trellis(spec)
        out[spec[0] + "__next"] = np.asarray(tr.next_state_table, dtype=np.int64)
        out[spec[0] + "__out"] = np.asarray(tr.output_table, dtype=np.int64)
        out[spec[0] + "__kn"] = np.array([tr.k, tr.n, tr.total_memory, tr.number_states, tr.number_inputs])
    # bit helpers (test_utilities.py:12-13 + wrap quirk utilities.py:81-85)
    out["dec2bit_17_8"] = dec2bitarray(17, 8)
    out["dec2bit_17_12_5"] = dec2bitarray((17, 12), 5)
    out["dec2bit_133_7"] = dec2bitarray(133, 7)
    out["dec2bit_171_7"] = dec2bitarray(171, 7)
    save("trellis", **out)


def gen_conv_encode():
    rs = np.random.RandomState(101)
    out = {}
    mes = np.array((0, 0, 1, 0))
    for spec in trellis_specs():
        tr = make_trellis(spec)
        if 4 % tr.k == 0:
            out[spec[0] + "__mes_cont"] = np.asarray(conv_encode(mes, tr, "cont"))
        msg = rs.randint(0, 2, 60)
        out[spec[0] + "__msg"] = msg
        out[spec[0] + "__term"] = np.asarray(conv_encode(msg, tr))
        out[spec[0] + "__cont"] = np.asarray(conv_encode(msg, tr, "cont"))
    # puncturing / depuncturing (Wifi80211 vectors wifi80211.py:75-89)
    for nm, pv in (("p23", [1, 1, 1, 0]), ("p34", [1, 1, 1, 0, 0, 1]), ("p56", [1, 1, 1, 0, 0, 1, 1, 0, 0, 1])):
        pv = np.array(pv)
        msg = rs.randint(0, 2, 120)
        pu = puncturing(msg, pv)
        out["punct_" + nm + "__vec"] = pv
        out["punct_" + nm + "__msg"] = msg
        out["punct_" + nm + "__punctured"] = pu
        out["punct_" + nm + "__depunctured"] = depuncturing(pu.astype(float) * 2 - 1, pv, 120)
    save("conv_encode", **out)


def gen_viterbi_small():
    """Grid: trellis x metric type x termination x tb_depth x noise. Mirrors test_convcode.py:133-178 variants."""
    rs = np.random.RandomState(17121996)
    out = {}
    names = []
    idx = 0
    for spec in trellis_specs():
        if spec[0] in ("k8_247_371",):
            lens = (48,)
        elif spec[0].startswith("k7") or spec[0].startswith("wifi"):
            lens = (60, 96)
        else:
            lens = (40, 100)
        tr = make_trellis(spec)
        for nbits in lens:
            nbits -= nbits % tr.k
            for term in ("term", "cont"):
                msg = rs.randint(0, 2, nbits)
                coded = conv_encode(msg, tr, term)
                for dtype in ("hard", "soft", "unquantized"):
                    for tb in (None, 15):
                        for noisy in (0, 1):
                            if dtype == "hard":
                                rx = coded.astype(float)
                                if noisy:
                                    flips = rs.rand(len(rx)) <= 0.06
                                    rx = np.where(flips, 1 - rx, rx)
                            elif dtype == "soft":
                                if noisy == 0:
                                    rx = 10.0 * coded - 5 + rs.randn(len(coded)) * 2
                                else:
                                    rx = 4.0 * coded - 2 + rs.randn(len(coded)) * 2.5
                            else:
                                rx = 2.0 * coded - 1 + rs.randn(len(coded)) * (0.3 if noisy == 0 else 0.9)
                            rx_in = rx.copy()
                            dec = viterbi_decode(rx.copy(), tr, tb, dtype)
                            key = "c%04d" % idx
                            names.append("%s|%s|%s|%s|%s|%d" % (key, spec[0], term, dtype, tb, noisy))
                            out[key + "__in"] = rx_in
                            out[key + "__out"] = np.asarray(dec, dtype=np.int64)
                            out[key + "__msg"] = msg
                            idx += 1
        # +-inf soft inputs (test_convcode.py:166-178)
        msg = rs.randint(0, 2, 40 - 40 % tr.k)
        for term in ("term", "cont"):
            coded = conv_encode(msg, tr, term)
            rx = np.where(coded == 1, np.inf, -np.inf)
            dec = viterbi_decode(rx.copy(), tr, 15, "soft")
            key = "c%04d" % idx
            names.append("%s|%s|%s|%s|%s|%d" % (key, spec[0], term, "soft", 15, 2))
            out[key + "__in"] = rx
            out[key + "__out"] = np.asarray(dec, dtype=np.int64)
            out[key + "__msg"] = msg
            idx += 1
    out["names"] = np.array(names)
    save("viterbi_small", **out)


def gen_viterbi_c1():
    """BASELINE config 1: K=3 [[5,7]], 64-bit blocks, hard decision over BSC(0.05)."""
    tr = Trellis(np.array([2]), np.array([[5, 7]]))
    B = 256
    msg = np.random.RandomState(1).randint(0, 2, (1000, 64))[:B]
    u = np.random.RandomState(2).rand(1000, 132)[:B]
    rx = np.empty((B, 132))
    dec = np.empty((B, 66), dtype=np.int64)
    for b in range(B):
        coded = conv_encode(msg[b], tr)
        rx[b] = coded ^ (u[b] <= 0.05)
        dec[b] = viterbi_decode(rx[b].copy(), tr, None, "hard")
    save("viterbi_c1", msg=msg, rx=rx, dec=dec)


def c2_inputs(B, ebn0_db, seed_msg=10, seed_noise=11):
    """BASELINE config 2 inputs through the reference modem."""
    tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
    md = QAMModem(4)
    msg = np.random.RandomState(seed_msg).randint(0, 2, (B, 1024))
    nrs = np.random.RandomState(seed_noise)
    N0 = md.Es / (0.5 * 2 * 10 ** (ebn0_db / 10.0))
    llr = np.empty((B, 2060))
    for b in range(B):
        c = conv_encode(msg[b], tr)
        s = md.modulate(c)
        y = s + np.sqrt(N0 / 2) * (nrs.randn(len(s)) + 1j * nrs.randn(len(s)))
        llr[b] = md.demodulate(y, "soft", N0)
    return tr, msg, llr, N0


def gen_viterbi_c2():
    out = {}
    for ebn0, B in ((3.0, 16), (1.0, 6)):
        tr, msg, llr, N0 = c2_inputs(B, ebn0)
        dec = np.empty((B, 1030), dtype=np.int64)
        t0 = time.time()
        for b in range(B):
            dec[b] = viterbi_decode(llr[b].copy(), tr, None, "soft")
        print("c2 ebn0=%.1f: %d codewords in %.1fs, BER %.2e" % (ebn0, B, time.time() - t0,
                                                                  np.mean(dec[:, :1024] != msg)))
        tag = "e%d" % int(ebn0)
        out[tag + "__msg"] = msg
        out[tag + "__llr"] = llr
        out[tag + "__dec"] = dec
        out[tag + "__N0"] = np.array(N0)
    save("viterbi_c2", **out)


def gen_map_turbo():
    out = {}
    names = []
    idx = 0
    rs = np.random.RandomState(20)
    specs = {s[0]: s for s in trellis_specs()}
    for tname in ("rsc_legacy_4", "rsc_legacy_8", "rsc_matrix_4", "t57"):
        tr = make_trellis(specs[tname])
        for N, ebn0 in ((64, 0.0), (200, 1.5), (256, 4.0)):
            msg = rs.randint(0, 2, N)
            coded = conv_encode(msg, tr, "cont")
            sys_b, par_b = coded[0::2], coded[1::2]
            nv = 1 / (2 * 0.5 * 10 ** (ebn0 / 10.0))
            sys_r = 2.0 * sys_b - 1 + np.sqrt(nv) * rs.randn(N)
            par_r = 2.0 * par_b - 1 + np.sqrt(nv) * rs.randn(N)
            for lint_kind in (0, 1):
                L_int = np.zeros(N) if lint_kind == 0 else rs.randn(N) * 2.0
                for mode in ("decode", "compute"):
                    L, bits = map_decode(sys_r.copy(), par_r.copy(), tr, nv, L_int.copy(), mode)
                    key = "m%03d" % idx
                    names.append("%s|%s|%d|%g|%d|%s" % (key, tname, N, nv, lint_kind, mode))
                    out[key + "__sys"] = sys_r
                    out[key + "__par"] = par_r
                    out[key + "__lint"] = L_int
                    out[key + "__nv"] = np.array(nv)
                    out[key + "__L"] = np.asarray(L)
                    out[key + "__bits"] = np.asarray(bits, dtype=np.int64)
                    idx += 1
    out["map_names"] = np.array(names)
    # turbo
    tnames = []
    idx = 0
    for tname, N, ebn0, iters, nblk in (("rsc_legacy_4", 128, 1.0, 4, 3), ("rsc_legacy_8", 128, 1.5, 3, 2),
                                        ("rsc_legacy_4", 1024, 1.5, 6, 2), ("rsc_legacy_4", 96, 3.0, 1, 1)):
        tr = make_trellis(specs[tname])
        il = RandInterlv(N, 1234)
        for b in range(nblk):
            msg = rs.randint(0, 2, N)
            s, p1, p2 = turbo_encode(msg, tr, tr, il)
            p2 = p2[:N]
            nv = 1 / (2 * (1.0 / 3) * 10 ** (ebn0 / 10.0))
            sr = 2.0 * s - 1 + np.sqrt(nv) * rs.randn(N)
            p1r = 2.0 * p1 - 1 + np.sqrt(nv) * rs.randn(N)
            p2r = 2.0 * p2 - 1 + np.sqrt(nv) * rs.randn(N)
            t0 = time.time()
            dec = turbo_decode(sr.copy(), p1r.copy(), p2r.copy(), tr, nv, iters, il)
            key = "t%03d" % idx
            tnames.append("%s|%s|%d|%g|%d" % (key, tname, N, nv, iters))
            out[key + "__msg"] = msg
            out[key + "__enc_s"] = np.asarray(s)
            out[key + "__enc_p1"] = np.asarray(p1)
            out[key + "__enc_p2"] = np.asarray(p2)
            out[key + "__sys"] = sr
            out[key + "__p1"] = p1r
            out[key + "__p2"] = p2r
            out[key + "__nv"] = np.array(nv)
            out[key + "__perm"] = np.asarray(il.p_array, dtype=np.int64)
            out[key + "__dec"] = np.asarray(dec, dtype=np.int64)
            print("turbo %s N=%d iters=%d: %.1fs, errors %d" % (tname, N, iters, time.time() - t0,
                                                                 int(np.sum(dec != msg))))
            idx += 1
    out["turbo_names"] = np.array(tnames)
    out["randinterlv_16_7"] = np.asarray(RandInterlv(16, 7).p_array, dtype=np.int64)
    save("map_turbo", **out)


# 802.11n-style QC-LDPC (1944,1296), Z=81 prototype (not shipped by the reference).
PROTO_1944_1296 = """
61 75  4 63 56  -  -  -  -  -  -  8  -  2 17 25  1  0  -  -  -  -  -  -
56 74 77 20  -  -  - 64 24  4 67  -  7  -  -  -  -  0  0  -  -  -  -  -
28 21 68 10  7 14 65  -  -  - 23  -  -  - 75  -  -  -  0  0  -  -  -  -
48 38 43 78 76  -  -  -  -  5 36  - 15 72  -  -  -  -  -  0  0  -  -  -
40  2 53 25  - 52 62  - 20  -  - 44  -  -  -  -  0  -  -  -  0  0  -  -
69 23 64 10 22  - 21  -  -  -  -  - 68 23 29  -  -  -  -  -  -  0  0  -
12  0 68 20 55 61  - 40  -  -  - 52  -  -  - 44  -  -  -  -  -  -  0  0
58  8 34 64 78  -  - 11 78 24  -  -  -  -  - 58  1  -  -  -  -  -  -  0
"""


def expand_qc(proto, Z):
    rows = [r.split() for r in proto.strip().splitlines()]
    mb, nb = len(rows), len(rows[0])
    H = np.zeros((mb * Z, nb * Z), dtype=np.int8)
    eye = np.eye(Z, dtype=np.int8)
    for i in range(mb):
        for j in range(nb):
            if rows[i][j] != "-":
                H[i * Z:(i + 1) * Z, j * Z:(j + 1) * Z] = np.roll(eye, int(rows[i][j]), axis=1)
    return H


def ldpc_param_arrays(p, prefix):
    keys = ("cnode_adj_list", "cnode_vnode_map", "vnode_adj_list", "vnode_cnode_map", "cnode_deg_list",
            "vnode_deg_list")
    d = {prefix + "__" + k: np.asarray(p[k]) for k in keys}
    d[prefix + "__dims"] = np.array([p["n_vnodes"], p["n_cnodes"], p["max_vnode_deg"], p["max_cnode_deg"]])
    return d


def gen_ldpc():
    out = {}
    names = []
    rs = np.random.RandomState(31)
    design_dir = os.path.join(REF, "commpy/channelcoding/designs/ldpc")
    # authored 802.11n-style design file lives in OUR package (reference text format ldpc.py:51-107)
    own = os.path.join(REPO, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt")
    if not os.path.exists(own):
        os.makedirs(os.path.dirname(own), exist_ok=True)
        write_ldpc_params(expand_qc(PROTO_1944_1296, 81), own)
    codes = {
        "gallager96": (os.path.join(design_dir, "gallager/96.33.964.txt"), 0.5),
        "gallager96b": (os.path.join(design_dir, "gallager/96.3.963.txt"), 0.5),
        "wimax960": (os.path.join(design_dir, "wimax/960.720.a.txt"), 0.75),
        "wimax1440": (os.path.join(design_dir, "wimax/1440.720.txt"), 0.5),
        "n1944": (own, 2.0 / 3),
    }
    idx = 0
    for cname, (path, rate) in codes.items():
        p = get_ldpc_code_params(path, True)
        out.update(ldpc_param_arrays(p, cname))
        n = p["n_vnodes"]
        plan = {
            "gallager96": [(1, 2.0, 100), (2, 2.5, 100), (3, 1.0, 7), (1, 6.0, 0), (2, 2.0, 1)],
            "gallager96b": [(2, 2.5, 20)],
            "wimax960": [(1, 3.5, 50), (2, 4.5, 5)],
            "wimax1440": [(2, 2.0, 50), (1, 1.0, 8), (1, 3.0, 2)],
            "n1944": [(1, 3.0, 50), (2, 3.6, 20)],
        }[cname]
        for (nblk, ebn0, iters) in plan:
            sigma = 1 / np.sqrt(10 ** (ebn0 / 10.0) * rate * 2)
            rx = 1.0 + sigma * rs.randn(n * nblk)          # all-zero codeword, BPSK 1-2c (test_ldpc.py:49-53)
            llr = 2.0 * rx / sigma ** 2
            for alg in ("SPA", "MSA"):
                t0 = time.time()
                dec, oll = ldpc_bp_decode(llr.copy(), p, alg, iters)
                key = "l%03d" % idx
                names.append("%s|%s|%d|%s|%d" % (key, cname, nblk, alg, iters))
                out[key + "__llr"] = llr
                out[key + "__dec"] = np.asarray(dec)
                out[key + "__out"] = np.asarray(oll)
                print("ldpc %s nblk=%d ebn0=%.1f %s iters=%d: %.2fs errs=%d" % (
                    cname, nblk, ebn0, alg, iters, time.time() - t0, int(np.sum(dec))))
                idx += 1
    # noiseless encode -> decode (test_ldpc.py:77-106) on a WiMax code, non-zero codeword
    p = get_ldpc_code_params(codes["wimax1440"][0], True)
    msg = rs.randint(0, 2, 1450)
    coded = triang_ldpc_systematic_encode(msg, p)
    out["enc1440__msg"] = msg
    out["enc1440__coded"] = np.asarray(coded)
    sym = np.where(coded == 1, -1.0, 1.0).reshape(-1, order="F")
    for alg in ("SPA", "MSA"):
        dec, oll = ldpc_bp_decode(sym.copy(), p, alg, 10)
        out["enc1440__dec_" + alg] = np.asarray(dec)
        out["enc1440__out_" + alg] = np.asarray(oll)
    # clipping case: huge LLRs
    pg = get_ldpc_code_params(codes["gallager96"][0], True)
    llr = rs.randn(96) * 400 + 300
    for alg in ("SPA", "MSA"):
        dec, oll = ldpc_bp_decode(llr.copy(), pg, alg, 10)
        key = "l%03d" % idx
        names.append("%s|%s|%d|%s|%d" % (key, "gallager96", 1, alg, 10))
        out[key + "__llr"] = llr
        out[key + "__dec"] = np.asarray(dec)
        out[key + "__out"] = np.asarray(oll)
        idx += 1
    out["names"] = np.array(names)
    save("ldpc", **out)


def gen_demod():
    out = {}
    names = []
    rs = np.random.RandomState(41)
    modems = {
        "qam4": QAMModem(4), "qam16": QAMModem(16), "qam64": QAMModem(64), "qam256": QAMModem(256),
        "psk2": PSKModem(2), "psk4": PSKModem(4), "psk8": PSKModem(8), "psk16": PSKModem(16),
        "custom4": Modem([1 + 1j, -1.2 + 0.8j, 0.3 - 1j, -1 - 1.5j]),
        "custom8_nogray": Modem(np.exp(1j * np.arange(8) * 2 * np.pi / 8) * np.array([1, 2, 1, 2, 1, 2, 1, 2]),
                                reorder_as_gray=False),
    }
    # custom8_nogray bypasses the constellation setter in the reference ctor path? (modulation.py:76-77 uses it)
    idx = 0
    for mname, md in modems.items():
        out[mname + "__const"] = np.asarray(md.constellation, dtype=complex)
        out[mname + "__Es"] = np.array(md.Es)
        nb = md.num_bits_symbol
        nsym = 24 if md.m >= 64 else 48
        bits = rs.randint(0, 2, nsym * nb)
        s = md.modulate(bits)
        out[mname + "__bits"] = bits
        out[mname + "__sym"] = np.asarray(s, dtype=complex)
        for snr_db in (3.0, 12.0):
            N0 = md.Es / 10 ** (snr_db / 10.0)
            y = s + np.sqrt(N0 / 2) * (rs.randn(nsym) + 1j * rs.randn(nsym))
            with np.errstate(all="ignore"):
                soft = md.demodulate(y, "soft", N0)
            hard = md.demodulate(y, "hard")
            key = "d%03d" % idx
            names.append("%s|%s|%g" % (key, mname, N0))
            out[key + "__y"] = y
            out[key + "__N0"] = np.array(N0)
            out[key + "__soft"] = np.asarray(soft)
            out[key + "__hard"] = np.asarray(hard)
            idx += 1
    out["names"] = np.array(names)
    save("demod", **out)


GENS = {
    "trellis": gen_trellis, "conv_encode": gen_conv_encode, "viterbi_small": gen_viterbi_small,
    "viterbi_c1": gen_viterbi_c1, "viterbi_c2": gen_viterbi_c2, "map_turbo": gen_map_turbo,
    "ldpc": gen_ldpc, "demod": gen_demod,
}




WIFI_RUNS = ((1, [3.0, 5.0, 7.0], 64), (5, [15.0, 18.0, 21.0], 64), (3, [9.0, 12.0], 64))


def _wifi_one(args):
    """One seeded sweep of the live reference's Wifi80211.link_performance; also records what the reference's receive
    chain saw (the channel output y of every transmission, in order) so that a test can replay the receiver alone."""
    os.environ["OMP_NUM_THREADS"] = "1"
    gname, gm, mcs, snrs, tx = args
    from commpy.wifi80211 import Wifi80211
    from commpy.channels import SISOFlatChannel
    Wifi80211.generator_matrix = gm
    np.random.seed(2024 + mcs)
    w = Wifi80211(mcs)
    ch = SISOFlatChannel(fading_param=(1 + 0j, 0j))
    t0 = time.time()
    bers, bes, ces, ncs = w.link_performance(ch, np.array(snrs), tx, 1, 600, stop_on_surpass_error=False)
    print("wifi %s mcs=%d snrs=%s -> BER %s (%.0fs)" % (gname, mcs, snrs, bers, time.time() - t0), flush=True)
    return gname, mcs, snrs, tx, np.asarray(bers), np.asarray(bes), np.asarray(ces), np.asarray(ncs)


def gen_wifi():
    """Wifi80211 link BER points (config 5 semantics, wifi80211.py:132-216 + links.py:155-267) from the live
    reference: decimal generators as shipped (quirk B1, catastrophic code) and the intended octal ones
    (class attribute overridden).  Every sweep runs under ``np.random.seed(2024 + mcs)`` with the argument list
    ``(channel, snrs, tx, 1, 600, stop_on_surpass_error=False)``, so the per-transmission error counts ``bes`` are a
    DETERMINISTIC fixture for any implementation that consumes NumPy's global stream in the reference's order
    (tests/test_wifi_gpu.py::test_wifi80211_error_counts_equal_the_reference) and a statistical one for the batched
    GPU path.  Round 6: 64 transmissions per point (was 12-30); the six sweeps run in parallel processes, each seeded
    on its own, so the result equals the sequential run's."""
    import multiprocessing as mp
    jobs = [(gname, gm, mcs, snrs, tx)
            for gname, gm in (("decimal", np.array((133, 171), ndmin=2)), ("octal", np.array((0o133, 0o171), ndmin=2)))
            for mcs, snrs, tx in WIFI_RUNS]
    out = {}
    names = []
    with mp.Pool(min(6, os.cpu_count())) as pool:
        for gname, mcs, snrs, tx, bers, bes, ces, ncs in pool.map(_wifi_one, jobs, chunksize=1):
            key = "w_%s_mcs%d" % (gname, mcs)
            names.append(key)
            out[key + "__snrs"] = np.array(snrs)
            out[key + "__ber"] = bers
            out[key + "__bes"] = bes
            out[key + "__ces"] = ces
            out[key + "__ncs"] = ncs
            out[key + "__tx"] = np.array(tx)
    out["names"] = np.array(names)
    save("wifi", **out)


GENS["wifi"] = gen_wifi

def _ber_one(args):
    os.environ["OMP_NUM_THREADS"] = "1"
    llr, msg = args
    tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
    dec = viterbi_decode(llr.copy(), tr, None, "soft")
    return int(np.sum(dec[:1024] != msg))


def gen_viterbi_ber():
    """BER-vs-Eb/N0 reference points for BASELINE config 2 (K=7 soft Viterbi, 1024-bit blocks, QPSK + AWGN):
    error counts of the live reference per Eb/N0 AND per codeword (``cw_errors``: the burstiness of Viterbi error
    events is then measured instead of assumed), for the statistical overlay of the GPU curve.  Round 6: 224 codewords
    (229 376 bits) per point, was 20-28."""
    import multiprocessing as mp
    ebn0s = [0.0, 1.0, 2.0, 3.0, 4.0]
    B = 224
    errs, nbits, per_cw = [], [], []
    with mp.Pool(min(6, os.cpu_count())) as pool:
        for i, e in enumerate(ebn0s):
            tr, msg, llr, N0 = c2_inputs(B, e, seed_msg=300 + i, seed_noise=400 + i)
            t0 = time.time()
            cw = np.array(pool.map(_ber_one, [(llr[b], msg[b]) for b in range(B)], chunksize=4))
            per_cw.append(cw)
            errs.append(int(cw.sum()))
            nbits.append(B * 1024)
            print("viterbi_ber Eb/N0=%.1f: %d errors in %d bits (%.0fs)" % (e, errs[-1], B * 1024, time.time() - t0), flush=True)
    save("viterbi_ber", ebn0=np.array(ebn0s), errors=np.array(errs), bits=np.array(nbits), cw_errors=np.stack(per_cw))


GENS["viterbi_ber"] = gen_viterbi_ber


# ----------------------------------------------------------------------------------------------
# Round-2 additions: the configs at (closer to) their real sizes, still through the LIVE reference.
# The slow decodes are spread over the container's cores (one codeword per task, the decoder call only).
# ----------------------------------------------------------------------------------------------
LLR_Q = 256.0          # config-2 LLRs are stored as int16 multiples of 1/256 (compact fixture, exact in float64)


def _c2x_one(args):
    os.environ["OMP_NUM_THREADS"] = "1"
    llr_q, = args
    tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
    return np.asarray(viterbi_decode(llr_q.astype(np.float64) / LLR_Q, tr, None, "soft"), dtype=np.uint8)


def gen_viterbi_c2x():
    """BASELINE config 2 at 256 codewords for each of Eb/N0 = 1, 3, 5 dB.  The inputs are the
    reference modem's soft LLRs rounded to multiples of 1/256 (stored as int16): every decoder under test reads
    exactly those float64 values, so the fixture stays small (about 2 MB) without weakening the comparison -- the
    branch metrics log(exp(r)+1) of such r are as arbitrary in their last bits as any."""
    import multiprocessing as mp
    out = {}
    B = 256
    with mp.Pool(os.cpu_count()) as pool:
        for ebn0 in (1.0, 3.0, 5.0):
            tr, msg, llr, N0 = c2_inputs(B, ebn0, seed_msg=110 + int(ebn0), seed_noise=210 + int(ebn0))
            q = np.clip(np.rint(llr * LLR_Q), -32767, 32767).astype(np.int16)
            t0 = time.time()
            dec = np.stack(pool.map(_c2x_one, [(q[b],) for b in range(B)], chunksize=4))
            print("c2x ebn0=%.1f: %d codewords in %.1fs, BER %.2e" % (ebn0, B, time.time() - t0,
                                                                       np.mean(dec[:, :1024] != msg)))
            tag = "e%d" % int(ebn0)
            out[tag + "__msg"] = np.packbits(msg.astype(np.uint8), axis=1)
            out[tag + "__llr_q"] = q
            out[tag + "__dec"] = np.packbits(dec, axis=1)
            out[tag + "__N0"] = np.array(N0)
    out["llr_scale"] = np.array(LLR_Q)
    save("viterbi_c2x", **out)


def _c3x_one(args):
    os.environ["OMP_NUM_THREADS"] = "1"
    sr, p1r, p2r, nv, iters, perm = args
    specs = {s[0]: s for s in trellis_specs()}
    tr = make_trellis(specs["rsc_legacy_4"])
    il = RandInterlv(len(perm), 1234)
    assert np.array_equal(il.p_array, perm)
    dec = turbo_decode(sr.copy(), p1r.copy(), p2r.copy(), tr, nv, iters, il)
    # the a-posteriori LLRs of the first MAP pass as well (map_decode is the float output the 1e-5 clause is about)
    L, _ = map_decode(sr.copy(), p1r.copy(), tr, nv, np.zeros(len(sr)), "compute")
    return np.asarray(dec, dtype=np.uint8), np.asarray(L)


def gen_turbo_c3x():
    """BASELINE config 3 shape (4-state legacy RSC, N = 1024, RandInterlv(1024, 1234), 6 iterations, Eb/N0 = 1.5 dB):
    48 noisy codewords through the live reference.  Received values are stored as float32 (the decoders read them
    widened to float64)."""
    import multiprocessing as mp
    specs = {s[0]: s for s in trellis_specs()}
    tr = make_trellis(specs["rsc_legacy_4"])
    N, B, iters = 1024, 48, 6
    il = RandInterlv(N, 1234)
    rs = np.random.RandomState(20)
    nv = 1 / (2 * (1.0 / 3) * 10 ** (1.5 / 10.0))
    msg = rs.randint(0, 2, (B, N))
    rx = np.empty((B, 3, N), dtype=np.float32)
    for b in range(B):
        s, p1, p2 = turbo_encode(msg[b], tr, tr, il)
        for j, a in enumerate((s, p1, p2[:N])):
            rx[b, j] = (2.0 * a - 1 + np.sqrt(nv) * rs.randn(N)).astype(np.float32)
    t0 = time.time()
    with mp.Pool(os.cpu_count()) as pool:
        res = pool.map(_c3x_one, [(rx[b, 0].astype(np.float64), rx[b, 1].astype(np.float64), rx[b, 2].astype(np.float64),
                                   nv, iters, np.asarray(il.p_array)) for b in range(B)], chunksize=1)
    dec = np.stack([r[0] for r in res])
    L1 = np.stack([r[1] for r in res])
    print("c3x: %d codewords in %.1fs, BER %.2e" % (B, time.time() - t0, np.mean(dec != msg)))
    save("turbo_c3x", msg=np.packbits(msg.astype(np.uint8), axis=1), rx=rx, nv=np.array(nv), iters=np.array(iters),
         perm=np.asarray(il.p_array, dtype=np.int32), dec=np.packbits(dec, axis=1), L_map1=L1)


def _c4x_one(args):
    os.environ["OMP_NUM_THREADS"] = "1"
    llr, alg, iters, path = args
    p = get_ldpc_code_params(path, True)
    dec, oll = ldpc_bp_decode(llr.copy(), p, alg, iters)
    return np.asarray(dec, dtype=np.int8), np.asarray(oll)


def gen_ldpc_c4x():
    """BASELINE config 4 chain at Eb/N0 = 8 dB and 9 dB: random codewords of the (1944,1296) code -> QAMModem(64) ->
    AWGN -> reference demodulate('soft') -> sign flip -> reference ldpc_bp_decode, SPA and MSA, 50 iterations, 12
    blocks per point and algorithm (a mix of converged and non-converged blocks at 8 dB).  The codewords come from
    this repo's host GF(2) generator (commpy_amd.devicelink.gf2_generator: the reference's real-valued inverse
    cannot encode this H); that is input generation only -- every stored output is the reference's."""
    import multiprocessing as mp
    sys.path.insert(0, REPO)
    from commpy_amd.devicelink import gf2_generator
    own = os.path.join(REPO, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt")
    p = get_ldpc_code_params(own, True)
    P = gf2_generator({k: v for k, v in p.items() if k not in ("generator_matrix",)})
    md = QAMModem(64)
    rs = np.random.RandomState(31)
    nblk, iters = 12, 50
    out, names, jobs = {}, [], []
    for ebn0 in (8.0, 9.0):
        msg = rs.randint(0, 2, (nblk, 1296)).astype(np.uint8)
        code = np.concatenate([msg, (msg.astype(np.int64) @ P.T.astype(np.int64) % 2).astype(np.uint8)], axis=1)
        H = p["parity_check_matrix"]
        assert not np.any((H @ code.T.astype(np.int64)) % 2), "not codewords"
        N0 = md.Es / ((2.0 / 3) * 6 * 10 ** (ebn0 / 10.0))
        s = md.modulate(code.reshape(-1))
        y = s + np.sqrt(N0 / 2) * (rs.randn(len(s)) + 1j * rs.randn(len(s)))
        with np.errstate(all="ignore"):
            llr = -md.demodulate(y, "soft", N0)              # LDPC convention: positive = bit 0 (quirk B6)
        tag = "e%d" % int(ebn0)
        out[tag + "__code"] = code
        out[tag + "__y"] = y
        out[tag + "__N0"] = np.array(N0)
        out[tag + "__llr"] = llr
        for alg in ("SPA", "MSA"):
            for b in range(nblk):
                jobs.append((tag, alg, b, (llr[b * 1944:(b + 1) * 1944].copy(), alg, iters, own)))
    t0 = time.time()
    with mp.Pool(os.cpu_count()) as pool:
        res = pool.map(_c4x_one, [j[3] for j in jobs], chunksize=1)
    for (tag, alg, b, _), (dec, oll) in zip(jobs, res):
        out.setdefault("%s__dec_%s" % (tag, alg), np.zeros((nblk, 1944), np.int8))[b] = dec
        out.setdefault("%s__out_%s" % (tag, alg), np.zeros((nblk, 1944)))[b] = oll
    for tag in ("e8", "e9"):
        for alg in ("SPA", "MSA"):
            d = out["%s__dec_%s" % (tag, alg)]
            ok = [int(np.array_equal(d[b], out[tag + "__code"][b].astype(np.int8))) for b in range(nblk)]
            print("c4x %s %s: decoded==sent per block %s" % (tag, alg, ok))
    print("c4x: %d reference decodes in %.1fs" % (len(jobs), time.time() - t0))
    out["iters"] = np.array(iters)
    save("ldpc_c4x", **out)


GENS["viterbi_c2x"] = gen_viterbi_c2x
GENS["turbo_c3x"] = gen_turbo_c3x
GENS["ldpc_c4x"] = gen_ldpc_c4x


def _c2u_one(args):
    os.environ["OMP_NUM_THREADS"] = "1"
    llr, = args
    tr = Trellis(np.array([6]), np.array([[0o133, 0o171]]))
    return np.asarray(viterbi_decode(llr.copy(), tr, None, "soft"), dtype=np.uint8)


def gen_viterbi_c2u():
    """BASELINE config 2, 256 codewords at Eb/N0 = 3 dB with the reference modem's float64 LLRs AS THEY ARE (VERDICT r02
    item 8: the 1/256-quantised inputs of viterbi_c2x make metric ties likelier than real inputs do)."""
    import multiprocessing as mp
    B = 256
    tr, msg, llr, N0 = c2_inputs(B, 3.0, seed_msg=313, seed_noise=413)
    t0 = time.time()
    with mp.Pool(os.cpu_count()) as pool:
        dec = np.stack(pool.map(_c2u_one, [(llr[b],) for b in range(B)], chunksize=4))
    print("c2u: %d codewords in %.1fs, BER %.2e" % (B, time.time() - t0, np.mean(dec[:, :1024] != msg)))
    save("viterbi_c2u", msg=np.packbits(msg.astype(np.uint8), axis=1), llr=llr, dec=np.packbits(dec, axis=1), N0=np.array(N0))


GENS["viterbi_c2u"] = gen_viterbi_c2u


def gen_abnormal():
    """Inputs outside the reference's representable range, through the LIVE reference: what it returns
    there -- NaN / +-inf LLRs, all-zero decisions after a NaN -- is part of its behaviour and pinned here.
      vit_*   viterbi_decode 'soft' with NaN among the LLRs (convcode.py:719 lets them through the clip)
      msa_*   ldpc_bp_decode 'MSA' with NaN LLRs (np.nan: one NaN sign)
      map_*   map_decode with symbol amplitudes 5 - 20 at sigma^2 <= 0.1, priors of |L| up to 200, and NaN / inf inputs
      tur_*   turbo_decode in the same regimes"""
    out, names = {}, []
    rs = np.random.RandomState(2024)
    # ---- Viterbi
    for tname, mem, g in (("t57", [2], [[5, 7]]), ("k7_133_171", [6], [[0o133, 0o171]])):
        tr = Trellis(np.array(mem), np.array(g))
        for i, steps in enumerate((40, 97)):
            B = 6
            rx = rs.randn(B, steps * 2) * 3
            rx[rs.rand(*rx.shape) < 0.01] = np.nan
            rx[0, rs.randint(rx.shape[1])] = np.nan
            rx[B - 1] = rs.randn(steps * 2) * 3                     # one codeword without a NaN
            key = "vit_%s_%d" % (tname, i)
            with np.errstate(all="ignore"):
                dec = np.stack([viterbi_decode(rx[b].copy(), tr, None, "soft") for b in range(B)])
            out[key + "__rx"], out[key + "__dec"] = rx, dec.astype(np.uint8)
            names.append(key)
    # ---- min-sum LDPC
    design_dir = os.path.join(REF, "commpy/channelcoding/designs/ldpc")
    for cname, path, iters in (("gallager96", "gallager/96.33.964.txt", 12), ("wimax960", "wimax/960.720.a.txt", 6)):
        p = get_ldpc_code_params(os.path.join(design_dir, path), True)
        n, B = p["n_vnodes"], 5
        llr = 2.0 * (1.0 + 0.8 * rs.randn(n * B)) / 0.64
        llr[0 * n + rs.randint(n)] = np.nan
        llr[2 * n + rs.randint(n, size=3)] = np.nan
        llr[3 * n:4 * n] = np.nan
        with np.errstate(all="ignore"):
            dec, oll = ldpc_bp_decode(llr.copy(), p, "MSA", iters)
        key = "msa_" + cname
        out[key + "__llr"], out[key + "__dec"], out[key + "__out"] = llr, np.asarray(dec).astype(np.int8), np.asarray(oll)
        out[key + "__iters"] = np.array(iters)
        names.append(key)
    # ---- sum-product with LLRs of exactly 0 (punctured bits): "Runtime Warnings are expected when llr = 0" (ldpc.py:214) -- the block
    # fills with NaN, and dec_word = signbit(out_llrs) / the early-termination test read the sign of those NaNs
    for cname, path in (("gallager96", "gallager/96.33.964.txt"), ("wimax960", "wimax/960.720.a.txt")):
        p = get_ldpc_code_params(os.path.join(design_dir, path), True)
        n = p["n_vnodes"]
        for iters in (1, 2, 3, 4, 6):
            B = 3
            llr = 2.0 * (1.0 + 0.8 * rs.randn(n * B)) / 0.64
            llr[0 * n + rs.randint(n)] = 0.0
            llr[1 * n + rs.randint(n, size=3)] = 0.0              # block 2 stays clean
            with np.errstate(all="ignore"):
                dec, oll = ldpc_bp_decode(llr.copy(), p, "SPA", iters)
            key = "spaz_%s_%d" % (cname, iters)
            out[key + "__llr"], out[key + "__dec"], out[key + "__out"] = llr, np.asarray(dec).astype(np.int8), np.asarray(oll)
            out[key + "__iters"] = np.array(iters)
            names.append(key)
    # ---- MAP
    tr4 = Trellis(np.array([2]), np.array([[1, 7]]), feedback=5, code_type="rsc")
    tr8 = Trellis(np.array([3]), np.array([[1, 0o15]]), feedback=0o13, code_type="rsc")
    idx = 0
    for tname, tr in (("rsc_legacy_4", tr4), ("rsc_legacy_8", tr8)):
        for amp in (1.0, 5.0, 20.0):
            for nv in (0.02, 0.1, 1.0):
                for lsc in (0.0, 5.0, 60.0):
                    B, N = 2, int(rs.randint(5, 90))
                    s_ = (rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.5) * amp
                    p_ = (rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.5) * amp
                    L = rs.randn(B, N) * lsc
                    Lo, bo = np.zeros((B, N)), np.zeros((B, N), np.uint8)
                    with np.errstate(all="ignore"):
                        for b in range(B):
                            Lo[b], bb = map_decode(s_[b].copy(), p_[b].copy(), tr, nv, L[b].copy(), "decode")
                            bo[b] = bb
                    key = "map_%03d" % idx
                    idx += 1
                    out[key + "__sys"], out[key + "__par"], out[key + "__Lint"] = s_, p_, L
                    out[key + "__L"], out[key + "__bits"], out[key + "__nv"] = Lo, bo, np.array(nv)
                    names.append(key + "|" + tname)
    # NaN / inf among the inputs
    B, N = 7, 60
    s_ = rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.7
    p_ = rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.7
    L = rs.randn(B, N) * 2
    s_[1, 17] = np.nan; p_[2, 3] = np.inf; L[3, 30] = np.inf; L[4, 31] = -np.inf; L[5, 0] = np.nan
    L[6, 10] = 800.0; L[6, 11] = -800.0
    Lo, bo = np.zeros((B, N)), np.zeros((B, N), np.uint8)
    with np.errstate(all="ignore"):
        for b in range(B):
            Lo[b], bb = map_decode(s_[b].copy(), p_[b].copy(), tr4, 0.5, L[b].copy(), "decode")
            bo[b] = bb
    key = "map_nonfinite"
    out[key + "__sys"], out[key + "__par"], out[key + "__Lint"] = s_, p_, L
    out[key + "__L"], out[key + "__bits"], out[key + "__nv"] = Lo, bo, np.array(0.5)
    names.append(key + "|rsc_legacy_4")
    # ---- turbo
    idx = 0
    for amp, nv, lsc in ((5.0, 0.02, 0.0), (20.0, 0.1, 0.0), (1.0, 0.1, 60.0), (5.0, 1.0, 5.0), (1.0, 0.004, 0.0)):
        B, N = 4, int(rs.randint(40, 160))
        il = RandInterlv(N, 77)
        s_, p1, p2 = ((rs.choice([-1.0, 1.0], size=(B, N)) + rs.randn(B, N) * 0.5) * amp for _ in range(3))
        L = rs.randn(B, N) * lsc if lsc else np.zeros((B, N))
        for iters in (1, 3):
            dec = np.zeros((B, N), np.uint8)
            with np.errstate(all="ignore"):
                for b in range(B):
                    dec[b] = turbo_decode(s_[b].copy(), p1[b].copy(), p2[b].copy(), tr4, nv, iters, il, L[b].copy() if lsc else None)
            key = "tur_%02d" % idx
            idx += 1
            out[key + "__sys"], out[key + "__p1"], out[key + "__p2"], out[key + "__Lint"] = s_, p1, p2, L
            out[key + "__perm"], out[key + "__dec"] = np.asarray(il.p_array), dec
            out[key + "__par"] = np.array([nv, iters, 1.0 if lsc else 0.0])
            names.append(key)
    out["names"] = np.array(names)
    nf = sum(int(np.sum(~np.isfinite(v))) for k, v in out.items() if k.endswith("__L"))
    print("abnormal: %d cases, %d non-finite reference LLRs" % (len(names), nf))
    save("abnormal", **out)


GENS["abnormal"] = gen_abnormal


def gen_viterbi_pairs():
    """Generator pairs beyond the fixture list -- the codes the table-driven and small-ring fused kernels of round 3 serve -- through
    the LIVE reference: memory 2..6, both end taps set, hard / soft / unquantized, default traceback depth (and depth 40 for K = 7)."""
    out, names = {}, []
    rs = np.random.RandomState(606)
    pairs = [(2, 7, 5), (3, 0o15, 0o17), (3, 0o13, 0o17), (4, 0o35, 0o23), (4, 0o27, 0o31), (5, 0o53, 0o75), (5, 0o61, 0o73),
             (6, 0o135, 0o147), (6, 0o165, 0o127), (6, 0o133, 0o171)]
    for mem, g0, g1 in pairs:
        tr = Trellis(np.array([mem]), np.array([[g0, g1]]))
        for dtype in ("hard", "soft", "unquantized"):
            for tb in ((None, 40) if (mem, g0) == (6, 0o133) else (None,)):
                B, nbits = 5, int(rs.randint(60, 140))
                msg = rs.randint(0, 2, (B, nbits))
                coded = np.stack([conv_encode(msg[b], tr) for b in range(B)]).astype(float)
                if dtype == "hard":
                    rx = np.where(rs.rand(*coded.shape) < 0.07, 1 - coded, coded)
                elif dtype == "soft":
                    rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 1.8
                else:
                    rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.8
                dec = np.stack([viterbi_decode(rx[b].copy(), tr, tb, dtype) for b in range(B)]).astype(np.uint8)
                key = "p%d_%o_%o_%s_%s" % (mem, g0, g1, dtype, tb)
                out[key + "__rx"], out[key + "__dec"] = rx, dec
                names.append(key)
    out["names"] = np.array(names)
    print("viterbi_pairs: %d cases" % len(names))
    save("viterbi_pairs", **out)


GENS["viterbi_pairs"] = gen_viterbi_pairs


def gen_general():
    """The reference's argument domain BEYOND what the specialised kernels serve (round 4): trellises of 256 and 512 states,
    k = 3, n = 8 / 10 outputs per step (NumPy's eight-accumulator add.reduce order in the branch metrics), traceback windows
    of hundreds of steps; map_decode / turbo_decode on 32- and 64-state RSC trellises; belief propagation on a Tanner graph
    with checks of 40 edges; 1024-point constellations.  All through the LIVE reference.

    K = 9 / n >= 8: the reference's own Trellis raises OverflowError for total memory >= 8 and for n >= 8 under NumPy 2 (an
    int8 overflow in bitarray2dec during its table construction), so those tables come from commpy_amd's host Trellis (pure
    Python, checked here against a plain shift-register model) and are handed to the reference's viterbi_decode /
    conv_encode as a duck-typed trellis object.  Every decode below is the reference's own."""
    sys.path.insert(0, REPO)
    from commpy_amd.channelcoding.convcode import Trellis as OurTrellis
    out, names = {}, []
    rs = np.random.RandomState(4040)

    class Duck:
        pass

    def duck(gs, mem):
        """feed-forward k = 1 code with generators `gs`: commpy_amd's tables, checked against a plain shift-register model"""
        t = OurTrellis(np.array([mem]), np.array([list(gs)]))
        # state = the last `mem` inputs, most recent first; 'MSB' polynomial format: bit w of g is the tap on D^w
        # (convcode.py:196-197), i.e. on bit mem - w of the register word (input, state)
        S, n = 1 << mem, len(gs)
        rev = [int(format(int(g), "0%db" % (mem + 1))[::-1], 2) for g in gs]
        for st in range(S):
            for u in (0, 1):
                reg = (u << mem) | st
                o = 0
                for gr in rev:
                    o = 2 * o + (bin(reg & gr).count("1") & 1)
                assert t.output_table[st, u] == o and t.next_state_table[st, u] == reg >> 1
        d = Duck()
        for a in ("k", "n", "total_memory", "number_states", "number_inputs", "code_type"):
            setattr(d, a, getattr(t, a))
        d.next_state_table = np.array(t.next_state_table)
        d.output_table = np.array(t.output_table)
        return d

    def viterbi_cases(tag, tr, nbits_list, tbs, B=3, types=("hard", "soft", "unquantized"), noise=1.0):
        out[tag + "__next"] = np.asarray(tr.next_state_table, dtype=np.int64)
        out[tag + "__outp"] = np.asarray(tr.output_table, dtype=np.int64)
        out[tag + "__kn"] = np.array([tr.k, tr.n, tr.total_memory])
        for dtype in types:
            for nbits, tb in zip(nbits_list, tbs):
                msg = rs.randint(0, 2, (B, nbits))
                coded = np.stack([conv_encode(msg[b], tr) for b in range(B)]).astype(float)
                if dtype == "hard":
                    rx = np.where(rs.rand(*coded.shape) < 0.06 * noise, 1 - coded, coded)
                elif dtype == "soft":
                    rx = 4.0 * coded - 2 + rs.randn(*coded.shape) * 1.7 * noise
                else:
                    rx = 2.0 * coded - 1 + rs.randn(*coded.shape) * 0.8 * noise
                t0 = time.time()
                dec = np.stack([viterbi_decode(rx[b].copy(), tr, tb, dtype) for b in range(B)]).astype(np.uint8)
                key = "%s|%s|%s|%d" % (tag, dtype, tb, nbits)
                out[key + "__rx"], out[key + "__dec"] = rx, dec
                names.append(key)
                print("  %s: %.1fs, residual errors %d" % (key, time.time() - t0, int(np.sum(dec[:, :nbits] != msg))))

    viterbi_cases("k9_561_753", duck((0o561, 0o753), 8), (96, 70), (None, 25))
    viterbi_cases("k10_1167_1545", duck((0o1167, 0o1545), 9), (60,), (None,), B=2, types=("soft",))
    viterbi_cases("k3n4", Trellis(np.array([1, 1, 1]), np.array([[1, 0, 0, 3], [0, 1, 0, 3], [0, 0, 1, 3]])), (120, 90), (None, 9))
    viterbi_cases("k3n5m2", Trellis(np.array([2, 1, 1]), np.array([[7, 0, 0, 5, 3], [0, 3, 0, 1, 2], [0, 0, 3, 2, 1]])), (90,), (None,))
    viterbi_cases("r1_8", duck((0o17, 0o15, 0o13, 0o11, 0o7, 0o5, 0o16, 0o12), 3), (80, 50), (None, 6), noise=3.0)
    viterbi_cases("r1_10", duck((7, 5, 6, 3, 7, 5, 4, 1, 2, 7), 2), (64,), (None,), noise=3.5)
    # a traceback window far beyond what the state-per-lane kernels hold in LDS (K = 7, depth 600 and depth > block)
    viterbi_cases("k7_tb", Trellis(np.array([6]), np.array([[0o133, 0o171]])), (700, 100), (600, 80), B=2, types=("soft", "hard"))
    out["vit_names"] = np.array(names)

    # ---- map_decode / turbo_decode beyond 16 states ----
    mnames, tnames = [], []
    for tag, mem, g, fb in (("rsc32", 5, 0o67, 0o45), ("rsc64", 6, 0o171, 0o133)):
        tr = Trellis(np.array([mem]), np.array([[1, g]]), fb, "rsc")
        out[tag + "__next"] = np.asarray(tr.next_state_table, dtype=np.int64)
        out[tag + "__outp"] = np.asarray(tr.output_table, dtype=np.int64)
        out[tag + "__kn"] = np.array([tr.k, tr.n, tr.total_memory])
        for N, ebn0 in ((48, 1.0), (100, 3.0)):
            msg = rs.randint(0, 2, N)
            coded = conv_encode(msg, tr, "cont")
            nv = 1 / (2 * 0.5 * 10 ** (ebn0 / 10.0))
            sys_r = 2.0 * coded[0::2] - 1 + np.sqrt(nv) * rs.randn(N)
            par_r = 2.0 * coded[1::2] - 1 + np.sqrt(nv) * rs.randn(N)
            for kind in (0, 1):
                L_int = np.zeros(N) if kind == 0 else rs.randn(N) * 2.0
                L, bits = map_decode(sys_r.copy(), par_r.copy(), tr, nv, L_int.copy(), "decode")
                key = "%s|%d|%d" % (tag, N, kind)
                out[key + "__sys"], out[key + "__par"], out[key + "__lint"] = sys_r, par_r, L_int
                out[key + "__nv"] = np.array(nv)
                out[key + "__L"], out[key + "__bits"] = np.asarray(L), np.asarray(bits, dtype=np.int64)
                mnames.append(key)
        N, iters = 64, 2
        il = RandInterlv(N, 99)
        msg = rs.randint(0, 2, N)
        s, p1, p2 = turbo_encode(msg, tr, tr, il)
        p2 = p2[:N]
        nv = 1 / (2 * (1.0 / 3) * 10 ** (2.0 / 10.0))
        sr, p1r, p2r = (2.0 * v[:N] - 1 + np.sqrt(nv) * rs.randn(N) for v in (s, p1, p2))
        dec = turbo_decode(sr.copy(), p1r.copy(), p2r.copy(), tr, nv, iters, il)
        key = "%s|turbo" % tag
        out[key + "__sys"], out[key + "__p1"], out[key + "__p2"] = sr, p1r, p2r
        out[key + "__nv"], out[key + "__iters"] = np.array(nv), np.array(iters)
        out[key + "__perm"] = np.asarray(il.p_array, dtype=np.int64)
        out[key + "__dec"] = np.asarray(dec, dtype=np.int64)
        tnames.append(key)
        print("  %s: turbo errors %d" % (key, int(np.sum(dec != msg))))
    out["map_names"], out["turbo_names"] = np.array(mnames), np.array(tnames)

    # ---- LDPC: checks of 40 edges (n = 400, 40 checks, column weight 4) ----
    import scipy.sparse as sp
    n_v, n_c, wc = 400, 40, 4
    H = np.zeros((n_c, n_v), np.int8)
    for v in range(n_v):
        H[(np.arange(wc) * 10 + v * 7 + v // 40) % n_c, v] = 1
    assert H.sum(1).min() == 40 and H.sum(1).max() == 40 and H.sum(0).min() == wc
    out["ldpc_H"] = H
    lnames = []
    for alg in ("MSA", "SPA"):
        for sigma, nblk, its in ((0.45, 2, 12), (0.7, 3, 8)):
            params = {"n_vnodes": n_v, "n_cnodes": n_c, "parity_check_matrix": sp.csc_matrix(H)}
            tx = np.ones(n_v * nblk)                         # all-zero codeword, BPSK +1, LLR = 2 y / sigma^2 (positive = bit 0)
            llr = 2 * (tx + sigma * rs.randn(n_v * nblk)) / sigma ** 2
            dec, ol = ldpc_bp_decode(llr.copy(), params, alg, its)
            key = "ldpc|%s|%g|%d" % (alg, sigma, its)
            out[key + "__llr"], out[key + "__dec"], out[key + "__out"] = llr, np.asarray(dec), np.asarray(ol)
            lnames.append(key)
            print("  %s: bit errors %d" % (key, int(np.asarray(dec).sum())))
    out["ldpc_names"] = np.array(lnames)

    # ---- 1024-QAM (separable but above the LDS tables) and a 512-point custom constellation ----
    dnames = []
    for tag, modem in (("qam1024", QAMModem(1024)), ("custom512", None)):
        if modem is None:
            pts = (rs.randn(512) + 1j * rs.randn(512)) * 3
            modem = Modem(pts)
        cst = np.asarray(modem.constellation)
        nsym = 6
        tx = cst[rs.randint(0, len(cst), nsym)]
        nvar = 0.5 if tag == "qam1024" else 0.05
        y = tx + np.sqrt(nvar / 2) * (rs.randn(nsym) + 1j * rs.randn(nsym))
        out[tag + "__cst"], out[tag + "__y"], out[tag + "__nv"] = cst, y, np.array(nvar)
        out[tag + "__soft"] = modem.demodulate(y, "soft", nvar)
        out[tag + "__hard"] = modem.demodulate(y, "hard")
        dnames.append(tag)
    out["demod_names"] = np.array(dnames)
    print("general: %d viterbi, %d map, %d turbo, %d ldpc, %d demod cases" % (len(names), len(mnames), len(tnames), len(lnames), len(dnames)))
    save("general", **out)


GENS["general"] = gen_general


def gen_ldpc_c4y():
    """Round 4, for the sum-product tolerance table (profiles/r04_spa_tolerance.*): the BASELINE config-4 chain at Eb/N0 = 8,
    9 and 10 dB, 24 blocks each, sum-product only, 50 iterations, through the reference.  Stored: the reference demodulator's
    LLRs (the decoder input), the reference's out_llrs / dec_word, the sent codewords."""
    import multiprocessing as mp
    sys.path.insert(0, REPO)
    from commpy_amd.devicelink import gf2_generator
    own = os.path.join(REPO, "commpy_amd/channelcoding/designs/ldpc/ieee80211n/1944.1296.txt")
    p = get_ldpc_code_params(own, True)
    P = gf2_generator({k: v for k, v in p.items() if k not in ("generator_matrix",)})
    md = QAMModem(64)
    rs = np.random.RandomState(4131)
    nblk, iters = 24, 50
    out, jobs = {}, []
    for ebn0 in (8.0, 9.0, 10.0):
        msg = rs.randint(0, 2, (nblk, 1296)).astype(np.uint8)
        code = np.concatenate([msg, (msg.astype(np.int64) @ P.T.astype(np.int64) % 2).astype(np.uint8)], axis=1)
        N0 = md.Es / ((2.0 / 3) * 6 * 10 ** (ebn0 / 10.0))
        s = md.modulate(code.reshape(-1))
        y = s + np.sqrt(N0 / 2) * (rs.randn(len(s)) + 1j * rs.randn(len(s)))
        with np.errstate(all="ignore"):
            llr = -md.demodulate(y, "soft", N0)
        tag = "e%d" % int(ebn0)
        out[tag + "__code"] = code
        out[tag + "__llr"] = llr.reshape(nblk, 1944)
        for b in range(nblk):
            jobs.append((tag, b, (llr[b * 1944:(b + 1) * 1944].copy(), "SPA", iters, own)))
    t0 = time.time()
    with mp.Pool(os.cpu_count()) as pool:
        res = pool.map(_c4x_one, [j[2] for j in jobs], chunksize=1)
    for (tag, b, _), (dec, oll) in zip(jobs, res):
        out.setdefault(tag + "__dec", np.zeros((nblk, 1944), np.int8))[b] = dec
        out.setdefault(tag + "__out", np.zeros((nblk, 1944)))[b] = oll
    for tag in ("e8", "e9", "e10"):
        ok = [int(np.array_equal(out[tag + "__dec"][b], out[tag + "__code"][b].astype(np.int8))) for b in range(nblk)]
        print("c4y %s: decoded==sent per block %s, max |out| %.1f" % (tag, ok, np.abs(out[tag + "__out"]).max()))
    print("c4y: %d reference decodes in %.1fs" % (len(jobs), time.time() - t0))
    out["iters"] = np.array(iters)
    save("ldpc_c4y", **out)


GENS["ldpc_c4y"] = gen_ldpc_c4y

# ----------------------------------------------------------------------------------------------
# Host channel helpers (round 5): awgn / bsc / bec of commpy/channels.py:630-708 under a seeded global NumPy generator
def gen_channels():
    from commpy.channels import awgn, bec, bsc
    rs = np.random.RandomState(77)
    out = {}
    xr = rs.randn(257)
    xc = rs.randn(131) + 1j * rs.randn(131)
    bits = rs.randint(0, 2, 500)
    for tag, x, snr, rate in (("real", xr, 4.5, 0.5), ("cplx", xc, 11.0, 2.0 / 3), ("ones", np.ones(64), 0.0, 1.0)):
        np.random.seed(1234)
        out["awgn_%s__x" % tag], out["awgn_%s__par" % tag] = x, np.array([snr, rate])
        out["awgn_%s__y" % tag] = awgn(x, snr, rate)
    for p in (0.0, 0.07, 0.5, 1.0):
        np.random.seed(4321)
        out["bsc_%g" % p] = bsc(bits, p)
        np.random.seed(4321)
        out["bec_%g" % p] = bec(bits, p)
    out["bits"] = bits
    save("channels", **out)


GENS["channels"] = gen_channels

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    for nm, fn in GENS.items():
        if a.only and nm not in a.only.split(","):
            continue
        t0 = time.time()
        fn()
        print("== %s done in %.1fs" % (nm, time.time() - t0))
