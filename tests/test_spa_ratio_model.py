"""Host model of the ratio-domain sum-product row (scripts/micro/spa_ratio_emul.py: the arithmetic of ldpc_resident_ratio_kernel in NumPy
float64) against the live-reference blocks of tests/golden/ldpc_c4y.npz (ldpc.py:144-254 at 8 / 9 / 10 dB: 72 blocks): the reformulation -- likelihood
ratios instead of LLRs, no exp / log inside an iteration -- keeps dec_word, the oracle's iteration counts and the banded out_llrs contract.
CPU only; the kernel itself is compared with the log-domain row and the reference in the -m gpu tests."""
import os
import sys

import numpy as np
import pytest

import oracle
from helpers import golden, ldpc_params, spa_contract

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "micro"))


@pytest.mark.parametrize("tag", ["e8", "e9", "e10"])
def test_ratio_domain_model_meets_the_contract_on_reference_blocks(tag):
    from spa_ratio_emul import decode
    p = ldpc_params("n1944")
    ec, ev = oracle.ldpc_edges(p)
    g = golden("ldpc_c4y")
    llr = g[tag + "__llr"]
    stats = dict(rows=0, near=0, blocks_it=0, blocks_near=0, slow_vars=0, vars=0)
    dec, out, its = decode(llr.copy(), ec, ev, int(p["n_cnodes"]), int(p["n_vnodes"]), int(g["iters"]), stats)
    _, _, io = oracle.ldpc_bp_decode(llr.reshape(-1).copy(), p, "SPA", int(g["iters"]), True)
    assert np.array_equal(dec, g[tag + "__dec"])
    assert np.array_equal(its, io)
    spa_contract(out, g[tag + "__out"], "ratio-domain model, " + tag)
    assert 0 < stats["near"] < 0.01 * stats["rows"]                 # rows near saturation exist and are rare at a correct LLR scale
