"""A seeded slice of the randomised differential run (scripts/fuzz_gpu.py) inside `-m gpu`: every driver run of the GPU
tests is also ~20 s of fuzzing against the CPU oracle -- Viterbi over every kernel path (fused, two-kernel, state-per-lane,
table-driven, small-ring, general), LDPC min-sum / sum-product on random Tanner graphs, MAP, turbo, PSK / QAM demodulation,
abnormal inputs included.  The fuzzer found real parity gaps in rounds 2 and 3 (near-underflow demodulation, NaN signs of a
sum-product decode with zero LLRs); longer runs with open-ended seeds stay a script."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [20260925, 4])
def test_fuzz_slice(gpu, seed, capsys):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import fuzz_gpu
    rc = fuzz_gpu.main(["--seconds", "10", "--seed", str(seed)])
    out = capsys.readouterr().out
    assert rc == 0, out[-3000:]
    assert "failures: 0" in out
    cases = out[out.rindex("cases:"):]
    print(cases)
