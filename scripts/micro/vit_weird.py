"""Viterbi on inputs nobody should send: non-binary 'hard' values, +-inf / NaN / 1e200 in 'unquantized', +-inf / NaN / +-500 / +-0 in
'soft', five trellises (k = 1 and 2, 4 to 128 states), all kernel paths, against the oracle (= the reference, checked on these cases).
A smaller draw of the same cases is part of the suite: tests/test_viterbi_gpu.py::test_abnormal_inputs_all_types_vs_oracle."""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle
from helpers import make_trellis
from commpy_amd import _lib
from commpy_amd.channelcoding import viterbi_decode
rs = np.random.RandomState(0)
bad = 0; n = 0
for name in ("k7_133_171", "t57", "k2_default", "rsc_legacy_8", "k8_247_371"):
    tr = make_trellis(name)
    for trial in range(60):
        B, steps = int(rs.choice([1, 5, 64, 70])), int(rs.randint(20, 150))
        length = steps * tr.n
        for dtype in ("hard", "unquantized", "soft"):
            if dtype == "hard":
                rx = rs.choice([0.0, 1.0, 2.0, -1.0, 0.5, 1.9, 3.0, -0.3], size=(B, length), p=[.4, .4, .04, .04, .03, .03, .03, .03])
            elif dtype == "unquantized":
                rx = rs.choice([-1.0, 1.0], size=(B, length)) + rs.randn(B, length) * 0.6
                for v in (np.inf, -np.inf, np.nan, 1e200, -1e200, 0.0):
                    rx[rs.rand(B, length) < 0.004] = v
            else:
                rx = rs.randn(B, length) * 4
                for v in (np.inf, -np.inf, np.nan, 1e200, 499.99999, -500.0, 0.0, -0.0):
                    rx[rs.rand(B, length) < 0.004] = v
            want = oracle.viterbi_decode(rx, tr, None, dtype)
            for path in ((None, "cw!", "cw2!", "wave") if name == "k7_133_171" else (None,)):
                _lib.viterbi_set_path(path)
                got = viterbi_decode(rx, tr, None, dtype)
                n += 1
                if not np.array_equal(got, want):
                    bad += 1
                    if bad <= 12:
                        print("FAIL", name, dtype, B, steps, path, int(np.sum(got != want)), flush=True)
            _lib.viterbi_set_path(None)
print("cases", n, "failures", bad)
