"""commpy_amd -- MI355X-native channel-decoding / demodulation engine with CommPy's Python API.

Scope: the batched decoding hot path of veeresht/CommPy 0.8.0 -- Viterbi,
BCJR/MAP + turbo, LDPC belief propagation and PSK/QAM hard/soft demodulation -- re-implemented as
hand-written HIP kernels for gfx950 behind a ctypes C-ABI (include/commpy_amd.h), plus the host-side
code descriptions either side of it (Trellis, interleavers, LDPC design files, constellations,
encoders).  No PyTorch, no Triton, no CPU fallback: the decoders raise if the HIP library or the
GPU is missing.

    from commpy_amd.channelcoding import Trellis, viterbi_decode, map_decode, turbo_decode, ldpc_bp_decode
    from commpy_amd.modulation import PSKModem, QAMModem
"""
__version__ = "0.3.0"

from commpy_amd import utilities  # noqa: F401



def set_precision(mode):
    """Precision mode of the engine: 'fp64-parity' (default: float64 in the reference's operation order, the mode every parity
    claim is made in) or 'fp32-fast' (float32 variants where a kernel has one; not bit-exact).  The switch is PROCESS-GLOBAL and affects
    two decoders: the large-batch Viterbi kernel (float32 path metrics) AND ``ldpc_bp_decode`` (float32 messages,
    hardware exp / log for sum-product, DESIGN.md 4.3: same decoded words on blocks that converge, NOT the 1e-5 LLR criterion).
    Enable it around the calls that should use it and reset it (``set_precision(None)``) afterwards.  Also settable through the
    environment variable CPX_PRECISION before the first call."""
    from commpy_amd import _lib
    _lib.set_precision(mode)


class precision:
    """``with commpy_amd.precision('fp32-fast'): ...`` -- the precision mode for the calls inside the block only; the previous mode is
    restored on the way out, exceptions included (round-4 answer to the process-global switch: enable it exactly around the decoder
    that should use it)."""

    def __init__(self, mode):
        self.mode = mode
        self._previous = None

    def __enter__(self):
        from commpy_amd import _lib
        self._previous = _lib.get_precision()
        _lib.set_precision(self.mode)
        return self

    def __exit__(self, *exc):
        from commpy_amd import _lib
        _lib.set_precision(self._previous)
        return False


__all__ = ["channelcoding", "modulation", "utilities", "parallel", "set_precision", "precision"]
