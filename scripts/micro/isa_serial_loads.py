#!/usr/bin/env python3
"""Scan the gfx950 assembly of the library's kernels for loads the compiler waits for one at a time: a global / buffer load followed by
`s_waitcnt vmcnt(0)` within six instructions and no other load in between.  Round 5 found the tiled sum-product check pass that way
(22 of 22 loads of a row; 1.7 TB/s -> 4.3 TB/s once the row's operands were requested together).
    python scripts/micro/isa_serial_loads.py [file.hip ...]      (default: every csrc/*.hip; compiles with --save-temps into a temp dir)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "commpy_amd", "csrc")


def scan(path):
    lines = open(path).read().split("\n")
    stats, cur = {}, None
    for ln, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s*; @", l)
        if m:
            cur = m.group(1)
            stats[cur] = [0, 0]
            continue
        if cur is None:
            continue
        t = l.strip()
        if t.startswith("s_endpgm"):
            cur = None
            continue
        if t.startswith("global_load") or t.startswith("buffer_load"):
            stats[cur][0] += 1
            k, seen = ln + 1, 0
            while k < len(lines) and seen < 6:
                tt = lines[k].strip()
                k += 1
                if not tt or tt.startswith(";") or tt.startswith("."):
                    continue
                seen += 1
                if tt.startswith("global_load") or tt.startswith("buffer_load"):
                    break
                if tt.startswith("s_waitcnt") and "vmcnt(0)" in tt:
                    stats[cur][1] += 1
                    break
    return stats


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with tempfile.TemporaryDirectory() as d:
        for src in srcs:
            base = os.path.splitext(os.path.basename(src))[0]
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--save-temps",
                            "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", os.path.abspath(src), "-o", base + ".o"],
                           cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            for k, (loads, serial) in scan(os.path.join(d, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).items():
                if loads >= 3 and serial >= 2:
                    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                    print("%-14s %3d of %3d loads waited for alone   %s" % (base, serial, loads, name[:110]))


if __name__ == "__main__":
    main()
