#!/usr/bin/env python3
"""Per-trellis-step instruction budget of the headline Viterbi kernel, read from the code object the in-tree build produced.

    python scripts/viterbi_step_budget.py [--kernel "viterbi_cw_fused_kernel<6, 109u, 79u, 1, 28, false, double, 32, true>"] [--md out.md]

Disassembles commpy_amd/csrc/build/viterbi_cw.o (device part, gfx950), finds the kernel's step loop (the smallest backward branch that spans
a whole trellis step), counts the steps of one trip of it by the first-equal scans it contains (64 v_cmp_eq_f64 per step) and
sorts every instruction of the loop body into the phases of cw_step (csrc/viterbi_cw.hip) by opcode -- the table of DESIGN.md 4.1."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT = "viterbi_cw_fused_kernel<6, 109u, 79u, 1, 28, false, double, 32, true>"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fatbin"), os.path.join(d, "co")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co], check=True)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    return subprocess.run(["c++filt"], input=dis, capture_output=True, text=True).stdout


def kernel_body(dis, name):
    out, on = [], False
    for line in dis.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", line)
        if m:
            on = name in m.group(2)
            continue
        if on:
            m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
            if m:
                out.append((int(m.group(3), 16), re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1)), m.group(2) + " //" + m.group(4)))
    return out


def classify(op, args):
    if op in ("v_add_f64",):
        return "add-compare-select: v_add_f64 (path metric + branch metric)"
    if op == "v_cmp_lt_f64" or op.startswith("v_addc_co"):
        return "add-compare-select: v_cmp_lt_f64 + v_addc_co_u32 (decision bit)"
    if op == "v_min_f64":
        return "v_min_f64 (64 survivor selects + 63 of the minimum tree)"
    if op == "v_cmp_eq_f64" or op == "v_cndmask_b32":
        return "first-equal scan: v_cmp_eq_f64 + v_cndmask_b32"
    if op.startswith(("v_exp", "v_log", "v_frexp", "v_ldexp", "v_rndne", "v_fma_f64", "v_mul_f64", "v_cvt", "v_div", "v_rcp", "v_max_f64",
                      "v_fmac_f64", "v_cmp_class", "v_cmp_u_f64", "v_cmp_gt_f64", "v_cmp_ngt", "v_cmp_nlt", "v_cmp_le_f64", "v_cmp_ge_f64",
                      "v_cmp_neq", "v_trig", "v_med3")):
        return "LLR -> branch metrics (clip, exp, log, the four sums)"
    if op.startswith("ds_"):
        return "LDS (decision ring write, traceback reads, output tile)"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "memory (LLR loads, bit stores)"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "scalar (loop control, NaN mask, addresses)"
    if op.startswith(("v_lshl", "v_lshr", "v_and", "v_or", "v_bfe", "v_alignbit", "v_xor", "v_add_u32", "v_sub", "v_add_co", "v_lshrrev", "v_mov",
                      "v_ashr", "v_bfi", "v_mad", "v_mul_lo", "v_mul_u32", "v_readlane", "v_readfirstlane", "v_add3", "v_lshl_add", "v_and_or",
                      "v_lshl_or", "v_perm", "v_accvgpr", "v_cmp_", "v_subrev", "v_min_u32", "v_max_u32", "v_min_i32", "v_not")):
        return "traceback hops, addresses, moves (integer VALU)"
    return "other (" + op + ")"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=DEFAULT)
    ap.add_argument("--obj", default=os.path.join(ROOT, "commpy_amd", "csrc", "build", "viterbi_cw.o"))
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    body = kernel_body(disassemble(a.obj), a.kernel)
    if not body:
        sys.exit("kernel not found: " + a.kernel)
    addr = {ad: i for i, (ad, _, _) in enumerate(body)}
    best = None
    for i, (ad, op, args) in enumerate(body):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"\+0x([0-9a-f]+)>\s*$", args)
            if not m:
                continue
            # the operand is printed relative to the kernel symbol: resolve through the first instruction's address
            tgt = body[0][0] + int(m.group(1), 16)
            j = addr.get(tgt)
            # the step loop: the SMALLEST backward branch whose span holds at least one whole trellis step (64 first-equal compares);
            # the loop around it flushes the output tile every 96 steps
            if j is not None and j < i and sum(1 for _, o, _ in body[j:i + 1] if o == "v_cmp_eq_f64") >= 64 and \
                    (best is None or i - j < best[1] - best[0]):
                best = (j, i)
    if best is None:
        sys.exit("no loop found")
    loop = body[best[0]:best[1] + 1]
    steps = sum(1 for _, op, _ in loop if op == "v_cmp_eq_f64") // 64
    cnt = collections.Counter(classify(op, args) for _, op, args in loop)
    valu = sum(v for k, v in cnt.items() if not k.startswith(("LDS", "memory", "s_", "scalar")))
    lines = ["kernel: %s" % a.kernel,
             "hot loop: %d instructions per trip, %d trellis steps per trip (kernel: %d instructions)" % (len(loop), steps, len(body)), "",
             "| phase of cw_step | instructions per trellis step |", "|---|---|"]
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
        lines.append("| %s | %.1f |" % (k, v / steps))
    lines += ["| **all** | **%.1f** (vector ALU: %.1f) |" % (len(loop) / steps, valu / steps)]
    text = "\n".join(lines)
    print(text)
    if a.md:
        open(a.md, "w").write(text + "\n")


if __name__ == "__main__":
    main()
