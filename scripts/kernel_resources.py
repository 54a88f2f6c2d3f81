#!/usr/bin/env python3
"""Register / LDS / scratch usage of every kernel of one translation unit, one line per kernel.

    python scripts/kernel_resources.py commpy_amd/csrc/viterbi_cw.hip [substring ...]

Compiles the file for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints
name, VGPRs, AGPRs, SGPRs, spills, scratch bytes, occupancy.  Optional substrings filter the demangled names.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    filters = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "commpy_amd/csrc"), "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*\)$", "", n).replace("void ", "")
        if filters and not any(f in n for f in filters):
            continue
        print(f"{n:90s} V{r.get('VGPRs', '?'):>4} A{r.get('AGPRs', '?'):>3} S{r.get('TotalSGPRs', '?'):>4} "
              f"spillV {r.get('VGPRs Spill', '?'):>3} spillS {r.get('SGPRs Spill', '?'):>3} "
              f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>4} occ {r.get('Occupancy [waves/SIMD]', '?')}")


if __name__ == "__main__":
    main()
