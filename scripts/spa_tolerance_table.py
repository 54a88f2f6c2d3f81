#!/usr/bin/env python3
"""Sum-product tolerance table: where do `out_llrs` of ldpc_bp_decode(…, 'SPA', 50) differ from the REFERENCE's by more than
1e-5, by |LLR| band and by iterations-to-converge?  (the documented bound and the tested bound have to be
the same thing.)

Inputs: tests/golden/ldpc_c4y.npz -- the BASELINE config-4 chain at Eb/N0 = 8 / 9 / 10 dB, 24 blocks each, decoded by the live
reference (tests/golden/make_golden.py gen_ldpc_c4y).  Decoders compared with it:
  oracle        oracle/cpx_oracle.c (glibc tanh / atanh, the reference's operation order)          -- CPU, always
  engine ratio  the library's default: likelihood-ratio state, no exp / log inside an iteration     -- GPU (child process)
  engine log    CPX_LDPC_SPA=log: the log-domain row, one division per edge, exact-order redo near saturation (round 3)
  engine exact  CPX_LDPC_SPA=exact: every row in the reference's operation order

    python scripts/spa_tolerance_table.py --out gpurun_out/r04           (GPU box;  --cpu-only in the build container)

Writes <out>/spa_tolerance.json and .md.  The iteration count of a block is the oracle's (the reference does not return one);
dec_word and iteration counts of every decoder are compared exactly."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BANDS = [(0, 1), (1, 10), (10, 26), (26, 50), (50, 100), (100, 501)]
ITS = [(0, 5), (5, 10), (10, 20), (20, 49), (49, 50)]


def run_engine(llr, p):
    from commpy_amd.channelcoding import ldpc_bp_decode
    dec, out, its = ldpc_bp_decode(llr.reshape(-1).copy(), p, "SPA", 50, return_iterations=True)
    return dec.T.copy(), out.T.copy(), its


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--cpu-only", action="store_true")
    ap.add_argument("--child", default=None, help="internal: dump the engine's outputs for this mode to the given .npz")
    a = ap.parse_args()
    from helpers import golden, ldpc_params
    g = golden("ldpc_c4y")
    p = ldpc_params("n1944")
    tags = ("e8", "e9", "e10")
    if a.child:
        res = {}
        for t in tags:
            dec, out, its = run_engine(g[t + "__llr"], p)
            res[t + "__dec"], res[t + "__out"], res[t + "__its"] = dec, out, its
        np.savez(a.child, **res)
        return 0
    import oracle
    dec_or, out_or, its_or = {}, {}, {}
    for t in tags:
        d, o, i = oracle.ldpc_bp_decode(g[t + "__llr"].reshape(-1).copy(), p, "SPA", 50, True)
        dec_or[t], out_or[t], its_or[t] = d.T, o.T, i
    decoders = {"oracle": (dec_or, out_or, its_or)}
    if not a.cpu_only:
        for mode, label in (("", "ratio (default)"), ("log", "log-domain fast row"), ("exact", "exact")):
            tmp = "/tmp/spa_tol_%s.npz" % (mode or "ratio")
            env = dict(os.environ)
            if mode:
                env["CPX_LDPC_SPA"] = mode
            else:
                env.pop("CPX_LDPC_SPA", None)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", tmp], env=env)
            z = np.load(tmp)
            decoders["engine " + label] = ({t: z[t + "__dec"] for t in tags}, {t: z[t + "__out"] for t in tags},
                                          {t: z[t + "__its"] for t in tags})
    table = {"fixture": "tests/golden/ldpc_c4y.npz (live reference, 3 x 24 blocks of the (1944,1296) code, 64-QAM chain, SPA, 50 iterations)",
             "bands": BANDS, "iteration_buckets": ITS, "decoders": {}}
    lines = ["# Sum-product `out_llrs` against the reference (tests/golden/ldpc_c4y.npz: 8 / 9 / 10 dB, 24 blocks each)", ""]
    for name, (dec, out, its) in decoders.items():
        ent = {"dec_word_mismatching_blocks": 0, "iteration_count_mismatches_vs_oracle": 0, "by_band": [], "by_iterations": [], "by_ebn0": {}}
        dev_all, mag_all, it_all = [], [], []
        for t in tags:
            ref_out, ref_dec = g[t + "__out"], g[t + "__dec"]
            ent["dec_word_mismatching_blocks"] += int(np.sum(np.any(dec[t] != ref_dec, axis=1)))
            ent["iteration_count_mismatches_vs_oracle"] += int(np.sum(np.asarray(its[t]) != its_or[t]))
            dev = np.abs(out[t] - ref_out)
            dev_all.append(dev)
            mag_all.append(np.abs(ref_out))
            it_all.append(np.repeat(its_or[t][:, None], dev.shape[1], axis=1))
            ent["by_ebn0"][t] = {"max_dev": float(dev.max()), "frac_beyond_1e-5": float(np.mean(dev > 1e-5)),
                                 "mean_iterations": float(np.mean(its_or[t])), "max_abs_llr": float(np.abs(ref_out).max())}
        dev, mag, itn = np.concatenate(dev_all).ravel(), np.concatenate(mag_all).ravel(), np.concatenate(it_all).ravel()
        for lo, hi in BANDS:
            m = (mag >= lo) & (mag < hi)
            ent["by_band"].append({"band": [lo, hi], "n": int(m.sum()), "frac_beyond_1e-5": float(np.mean(dev[m] > 1e-5)) if m.any() else None,
                                   "max_dev": float(dev[m].max()) if m.any() else None,
                                   "max_rel_dev": float((dev[m] / np.maximum(mag[m], 1e-300)).max()) if m.any() else None})
        for lo, hi in ITS:
            m = (itn > lo) & (itn <= hi)
            lo26 = m & (mag <= 26)
            ent["by_iterations"].append({"iterations": [lo + 1, hi], "n": int(m.sum()),
                                         "frac_beyond_1e-5": float(np.mean(dev[m] > 1e-5)) if m.any() else None,
                                         "max_dev": float(dev[m].max()) if m.any() else None,
                                         "max_dev_below_26": float(dev[lo26].max()) if lo26.any() else None})
        table["decoders"][name] = ent
        lines += ["## %s" % name, "",
                  "`dec_word` differs from the reference's in %d of 72 blocks; iteration counts differ from the oracle's in %d."
                  % (ent["dec_word_mismatching_blocks"], ent["iteration_count_mismatches_vs_oracle"]), "",
                  "| \\|LLR\\| band | values | beyond 1e-5 | max \\|dev\\| | max relative |", "|---|---|---|---|---|"]
        for r in ent["by_band"]:
            if r["n"]:
                lines.append("| [%g, %g) | %d | %.4f %% | %.3g | %.3g |" % (r["band"][0], r["band"][1], r["n"], 100 * r["frac_beyond_1e-5"],
                                                                          r["max_dev"], r["max_rel_dev"]))
        lines += ["", "| iterations of the block | values | beyond 1e-5 | max \\|dev\\| | max \\|dev\\| where \\|LLR\\| <= 26 |", "|---|---|---|---|---|"]
        for r in ent["by_iterations"]:
            if r["n"]:
                lines.append("| %d - %d | %d | %.4f %% | %.3g | %s |" % (r["iterations"][0], r["iterations"][1], r["n"], 100 * r["frac_beyond_1e-5"],
                                                                       r["max_dev"], "%.3g" % r["max_dev_below_26"] if r["max_dev_below_26"] is not None else "-"))
        lines += ["", "| Eb/N0 | mean iterations | max \\|LLR\\| | beyond 1e-5 | max \\|dev\\| |", "|---|---|---|---|---|"]
        for t in tags:
            r = ent["by_ebn0"][t]
            lines.append("| %s dB | %.1f | %.0f | %.4f %% | %.3g |" % (t[1:], r["mean_iterations"], r["max_abs_llr"], 100 * r["frac_beyond_1e-5"], r["max_dev"]))
        lines.append("")
    text = "\n".join(lines)
    print(text)
    if a.out:
        os.makedirs(a.out, exist_ok=True)
        json.dump(table, open(os.path.join(a.out, "spa_tolerance.json"), "w"), indent=1)
        open(os.path.join(a.out, "spa_tolerance.md"), "w").write(text + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
