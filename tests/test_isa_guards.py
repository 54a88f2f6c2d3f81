"""Build-time guards on the generated gfx950 code (CPU only: they disassemble the objects the in-tree build left behind).

The table-driven codeword-per-lane Viterbi step (csrc/viterbi_cw.hip, GEN) selects its branch metrics with VGPR INDEX MODE:
`s_set_gpr_idx_on` once per trellis step, `s_set_gpr_idx_idx` per butterfly pair, `s_set_gpr_idx_off` at the end -- with compiler-
generated code (the pipelined traceback hop, the register moves) in between.  Index mode lives in M0[7:0] and redirects the first
source operand of EVERY vector instruction, so the kernel is only correct as long as nothing else writes M0, or consumes it, inside
such a region (round-5 advisor finding).  The asm statements name m0 as clobbered; this test checks what the compiler really
emitted."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "commpy_amd", "csrc", "build", "viterbi_cw.o")
LLVM = "/opt/rocm/lib/llvm/bin"


def _disassemble(obj, tmp):
    fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "gfx950.co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], check=True)
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout


@pytest.mark.skipif(not (os.path.exists(OBJ) and os.path.isdir(LLVM) and shutil.which("c++filt")),
                    reason="needs the in-tree build's viterbi_cw.o and the ROCm LLVM tools")
def test_nothing_touches_m0_while_vgpr_index_mode_is_on(tmp_path):
    regions, longest, bad, kern, on = 0, 0, [], None, False
    span = 0
    for line in _disassemble(OBJ, str(tmp_path)).split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            assert not on, "index mode still on at the end of " + str(kern)
            kern = m.group(1)
            continue
        text = line.strip().split("//")[0].strip()
        if not text:
            continue
        op = text.split()[0]
        if op == "s_set_gpr_idx_on":
            assert not on, "nested s_set_gpr_idx_on in " + kern
            on, span = True, 0
            regions += 1
        elif op == "s_set_gpr_idx_off":
            on = False
            longest = max(longest, span)
        elif on:
            span += 1
            if op == "s_set_gpr_idx_idx":
                continue
            operands = [o.strip() for o in text[len(op):].split(",")]
            writes_m0 = bool(operands) and operands[0] == "m0"
            uses_m0 = op.startswith(("s_movrel", "v_movrel", "v_interp", "ds_gws", "s_sendmsg", "s_set_gpr_idx_mode")) or \
                text.endswith(" lds") or " lds " in text or "m0" in operands[1:]
            if writes_m0 or uses_m0 or op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                bad.append((kern, text))
    assert regions >= 8, regions                      # the table-driven kernels are there (several instantiations, unrolled steps)
    assert longest >= 100, longest                    # ... and compiler-generated code really runs inside the regions
    assert not bad, bad[:10]
