#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration on the GPU box: builds scripts/micro/fetch_calib.hip, runs it under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes, no other trace domain)
and writes <out>/fetch_calibration.json:  {"bytes": 2^30, "patterns": {"rd16": {"reported_bytes": ..., "factor": ...}, ...}}
with factor = known bytes / reported bytes (counter values are KiB).  scripts/collect_pmc.py --calibration <file> applies the
factor of the pattern named per kernel.

    python scripts/micro/fetch_calib.py --out gpurun_out/r04
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    exe = "/tmp/fetch_calib"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(HERE, "fetch_calib.hip")])
    env = dict(os.environ, TMPDIR="/tmp")
    known = 1 << 30
    res = {"bytes": known, "units": "reported_bytes = mean counter value per dispatch x 1024 (rocprofv3 reports KiB)", "patterns": {}}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = "/tmp/fetch_calib_" + counter
        shutil.rmtree(d, ignore_errors=True)
        subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", exe],
                       cwd="/tmp", env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
        acc = {}
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if r.get("Counter_Name") != counter:
                    continue
                k = r["Kernel_Name"].split("(")[0].strip()
                acc.setdefault(k, []).append(float(r.get("Counter_Value", 0) or 0))
        for k, vals in acc.items():
            want = counter == ("FETCH_SIZE" if k.startswith("rd") else "WRITE_SIZE")
            rep = sum(vals) / len(vals) * 1024
            e = res["patterns"].setdefault(k, {})
            e[counter.lower() + "_reported_bytes"] = rep
            if want:
                e["factor"] = known / rep if rep else None
        shutil.rmtree(d, ignore_errors=True)
    json.dump(res, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    sys.exit(main())
