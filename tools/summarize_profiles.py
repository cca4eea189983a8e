#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<round>_*: per-kernel time tables, PMC traffic
per kernel launch with the calibration factors measured on kernels of known byte count, and the per-entry-point traffic
bench.py reports as roofline.traffic.

  python tools/summarize_profiles.py final r01
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", name).strip()


def kernel_stats(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    if f:
        for r in csv.DictReader(open(f[0])):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6,
                         float(r["Percentage"])))
    return rows


def pmc(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.OrderedDict()
    if f:
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: (n, v / n * 1024.0) for k, (n, v) in acc.items()}  # KB -> bytes per launch


def bench_line(path):
    if os.path.exists(path):
        for line in open(path):
            if line.startswith("{"):
                return json.loads(line)
    return None


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    out = {}
    for name in ("stats256", "stats512", "statsslab"):
        rows = kernel_stats(os.path.join(src, name))
        if not rows:
            continue
        with open(os.path.join(dst, f"{rnd}_{name}_kernel_stats.csv"), "w") as fo:
            fo.write("kernel,calls,avg_us,total_ms,percent\n")
            for r in rows:
                fo.write(f"\"{r[0]}\",{r[1]},{r[2]:.2f},{r[3]:.3f},{r[4]:.2f}\n")
        b = bench_line(os.path.join(src, name + ".log"))
        if b:
            json.dump(b, open(os.path.join(dst, f"{rnd}_{name}_bench.json"), "w"), indent=1)
        out[name] = rows
    # ---- PMC traffic ----
    n = 256
    nxc, P = n // 2 + 1, 144
    cbytes = 8 * nxc * n * n  # one half-spectrum, valid columns
    calib = {}
    fp, wp = pmc(os.path.join(src, "pmcprobe_FETCH_SIZE"), "FETCH_SIZE"), pmc(os.path.join(src, "pmcprobe_WRITE_SIZE"), "WRITE_SIZE")
    known = {"k_probe_copy4": 8 * P * n * n, "k_probe_tile<256, 0>": cbytes, "k_probe_tile<256, 1>": cbytes}
    for k, true_b in known.items():
        if k in fp and k in wp:
            calib[k] = {"true_bytes_each_way": true_b, "FETCH_SIZE": fp[k][1], "WRITE_SIZE": wp[k][1],
                        "fetch_ratio": fp[k][1] / true_b, "write_ratio": wp[k][1] / true_b}
    f256, w256 = pmc(os.path.join(src, "pmc256_FETCH_SIZE"), "FETCH_SIZE"), pmc(os.path.join(src, "pmc256_WRITE_SIZE"), "WRITE_SIZE")
    # correction: FETCH_SIZE under-reports wide streaming reads (guide: exactly 1/2 for 16 B/lane); use the factor
    # measured on this build's own probes (8 B/lane tile loads and 16 B/lane flat loads)
    r8 = calib.get("k_probe_tile<256, 0>", {}).get("fetch_ratio")
    r16 = calib.get("k_probe_copy4", {}).get("fetch_ratio")
    table = []
    for k in f256:
        if k not in w256:
            continue
        table.append({"kernel": k, "launches": f256[k][0], "FETCH_SIZE_bytes": f256[k][1], "WRITE_SIZE_bytes": w256[k][1]})
    res = {"grid": [n, n, n], "calibration": calib, "fetch_ratio_8B_per_lane": r8, "fetch_ratio_16B_per_lane": r16,
           "per_kernel_launch": table}
    json.dump(res, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
