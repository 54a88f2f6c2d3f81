#!/usr/bin/env python3
"""rocprofv3 evidence for one workload: kernel-trace stats + separate PMC passes, summarised per kernel into JSON.

    python scripts/collect_pmc.py --out gpurun_out/r02 --name viterbi_c2 --match viterbi -- python bench.py --steps 3 ...

Runs the command once under `rocprofv3 --kernel-trace --stats` and once per counter group under
`rocprofv3 --pmc <group> --kernel-trace` (counters in their own runs, never mixed with other trace domains;
FETCH_SIZE and WRITE_SIZE in separate passes -- they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
Writes <out>/<name>_pmc.json: for every kernel whose name contains --match, the mean value per dispatch of every
counter, the mean duration, and derived figures:

  traffic_bytes_per_launch   FETCH_SIZE (KiB) x 1024 x fetch_scale + WRITE_SIZE (KiB) x 1024; --fetch-scale 2 (the default) is
                             the calibrated gfx950 correction: FETCH_SIZE reports HALF the bytes of every read pattern the
                             decoders use -- 16 / 8 / 4 / 1 byte per lane and 64-byte buffer-load segments, factors 1.98 - 2.00
                             on a known 1 GiB (scripts/micro/fetch_calib.py -> profiles/r04_fetch_calibration.json) --, WRITE_SIZE
                             is exact for 8- and 16-byte stores (1.00) and 4 % high for byte stores
  valu.busy_frac             SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) x 4 / (SIMDs in use x kernel cycles),
                             kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the XCDs)
  valu.insts_per_launch      SQ_INSTS_VALU
Raw rocprofv3 output directories are deleted; only the JSON and the stats CSV are kept (copy them to profiles/).
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

GROUPS = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
     "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT"],
    ["GRBM_GUI_ACTIVE"],
]
CLOCK_HZ = 2.4e9
SIMDS = 1024
N_XCD = 8


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--name", required=True)
    ap.add_argument("--match", default="")
    ap.add_argument("--fetch-scale", type=float, default=2.0)
    ap.add_argument("--batch", type=int, default=None, help="recorded in the JSON (bench.py checks it)")
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--groups", default=None, help='counter passes instead of the default ones: "A,B;C,D" = two passes')
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    kern = collections.defaultdict(dict)

    # ---- pass 0: kernel trace + stats ----
    d = os.path.join(out, a.name + "_trace")
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + cmd,
                   cwd="/tmp", env=env, timeout=a.timeout, stdout=open(os.path.join(out, a.name + "_trace.log"), "w"),
                   stderr=subprocess.STDOUT)
    # bench.py's own line from INSIDE the profiled run (round 5): its shader-clock probe and per-launch event times under rocprofv3
    profiled_line = None
    try:
        for ln in open(os.path.join(out, a.name + "_trace.log"), errors="replace"):
            if ln.startswith("{") and '"roofline"' in ln:
                j = json.loads(ln)
                profiled_line = {"ms_per_step": j.get("ms_per_step"), "kernel_ms_avg": j["roofline"].get("kernel_ms_avg"),
                                 "kernel_ms_median": j["roofline"].get("kernel_ms_median"), "clock": j["roofline"].get("clock")}
    except (OSError, ValueError, KeyError):
        pass
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, a.name + "_kernel_stats.csv"))
        for r in csv.DictReader(open(stats[0])):
            if a.match in r["Name"]:
                kern[short(r["Name"])]["duration_ns_avg"] = float(r["AverageNs"])
                kern[short(r["Name"])]["calls"] = int(r["Calls"])
    shutil.rmtree(d, ignore_errors=True)

    # ---- counter passes ----
    for grp in ([g.split(",") for g in a.groups.split(";")] if a.groups else GROUPS):
        tag = "_".join(grp)[:40]
        d = os.path.join(out, a.name + "_pmc_" + tag)
        shutil.rmtree(d, ignore_errors=True)
        subprocess.run(["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd,
                       cwd="/tmp", env=env, timeout=a.timeout,
                       stdout=open(os.path.join(out, a.name + "_pmc_" + tag + ".log"), "w"), stderr=subprocess.STDOUT)
        acc = collections.defaultdict(lambda: [0.0, 0])
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if a.match in r.get("Kernel_Name", ""):
                    k = (short(r["Kernel_Name"]), r["Counter_Name"])
                    acc[k][0] += float(r.get("Counter_Value", 0) or 0)
                    acc[k][1] += 1
        for (k, c), (tot, n) in acc.items():
            kern[k][c] = tot / max(n, 1)
            kern[k].setdefault("dispatches_counted", n)
        shutil.rmtree(d, ignore_errors=True)

    # ---- provenance: which code produced these numbers (bench.py refuses counters of another build) ----
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=root, capture_output=True, text=True, timeout=20).stdout.strip() or None
    except Exception:
        head = None
    if not head:                                       # the GPU box receives the tree without .git: scripts/run_gpu_round.sh
        try:                                           # stamps the commit (and whether the tree was dirty) into .git_head
            head = open(os.path.join(root, ".git_head")).read().strip() or None
        except OSError:
            head = None
    try:
        sys.path.insert(0, root)
        from commpy_amd import _lib
        build_id = _lib.build_id()
    except Exception:
        build_id = None

    # ---- derived ----
    res = {"name": a.name, "command": " ".join(cmd), "batch": a.batch, "fetch_scale": a.fetch_scale,
           "git_head": head, "build_id": build_id, "bench_line_under_kernel_trace": profiled_line,
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles "
                    "summed over waves; GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs",
           "kernels": {}}
    for k, v in kern.items():
        e = dict(v)
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            e["fetch_bytes_per_launch"] = v["FETCH_SIZE"] * 1024 * a.fetch_scale
            e["write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024
            e["traffic_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (32.2 M for a 1.85 ms kernel = 8 x 4.03 M cycles)
        cycles = (v.get("GRBM_GUI_ACTIVE", 0) / N_XCD) or (v.get("duration_ns_avg", 0) * 1e-9 * CLOCK_HZ)
        if v.get("GRBM_GUI_ACTIVE") and v.get("duration_ns_avg"):
            # the clock the kernel ran at under the profiler: cycles of the PMC pass over the duration of the kernel-trace pass
            e["sclk_mhz_under_profiler"] = v["GRBM_GUI_ACTIVE"] / N_XCD / (v["duration_ns_avg"] * 1e-9) * 1e-6
            e["shader_cycles_per_launch"] = v["GRBM_GUI_ACTIVE"] / N_XCD
        if cycles and "SQ_ACTIVE_INST_VALU" in v:
            simds = min(SIMDS, v.get("SQ_WAVES", SIMDS)) or SIMDS
            e["valu"] = {"insts_per_launch": v.get("SQ_INSTS_VALU"),
                         "active_quad_cycles_per_launch": v["SQ_ACTIVE_INST_VALU"],
                         "kernel_cycles": cycles, "simds_in_use": simds,
                         "busy_frac": v["SQ_ACTIVE_INST_VALU"] * 4.0 / (simds * cycles),
                         "cycles_per_valu_inst": v["SQ_ACTIVE_INST_VALU"] * 4.0 / v["SQ_INSTS_VALU"] if v.get("SQ_INSTS_VALU") else None,
                         "wait_any_frac": v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None,
                         "wait_inst_any_frac": v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None}
        res["kernels"][k] = e
    # the dominant kernel (longest total time) at the top level, in the form bench.py reads
    if res["kernels"]:
        dom = max(res["kernels"].items(), key=lambda kv: kv[1].get("duration_ns_avg", 0) * kv[1].get("calls", 1))
        res["kernel"] = dom[0]
        for key in ("traffic_bytes_per_launch", "fetch_bytes_per_launch", "write_bytes_per_launch", "valu", "duration_ns_avg",
                    "sclk_mhz_under_profiler", "shader_cycles_per_launch"):
            if key in dom[1]:
                res[key] = dom[1][key]
    json.dump(res, open(os.path.join(out, a.name + "_pmc.json"), "w"), indent=1)
    print(json.dumps({k: {c: (round(x, 4) if isinstance(x, float) else x) for c, x in v.items() if c != "valu"}
                      for k, v in res["kernels"].items()}, indent=1)[:6000])


if __name__ == "__main__":
    sys.exit(main())
