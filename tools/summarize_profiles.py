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
                         float(r["Percentage"]), float(r.get("MinNs", 0)) / 1e3, float(r.get("MaxNs", 0)) / 1e3))
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


def sq_counters(d, out_path):
    """issue-side picture per kernel from one SQ counter pass (sums over the chip, averaged over launches)"""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        a = acc.setdefault(short(r["Kernel_Name"]), collections.OrderedDict())
        c = a.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1
        c[1] += float(r["Counter_Value"])
    with open(out_path, "w") as fo:
        fo.write("# SQ counters per kernel launch, 256^3 config 3 (rocprofv3 --pmc SQ_* --kernel-trace, its own pass).\n"
                 "#   valu/inst  = SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY   share of issue activity that is VALU\n"
                 "#   wait/wave  = SQ_WAIT_ANY / SQ_WAVE_CYCLES                share of wave lifetime spent waiting on anything\n"
                 "#   winst/wave = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES           share waiting for an issue slot\n"
                 "#   inst/wave  = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES;  lds/inst = SQ_ACTIVE_INST_LDS / SQ_ACTIVE_INST_ANY\n")
        fo.write(f"{'kernel':42s} {'launches':>8s} {'waves':>7s} {'valu/inst':>9s} {'wait/wave':>9s} {'winst/wave':>10s} "
                 f"{'inst/wave':>9s} {'lds/inst':>8s}\n")
        for k, a in acc.items():
            v = {c: x[1] / x[0] for c, x in a.items()}
            if not all(c in v for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
                                        "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAVES")):
                continue
            n = list(a.values())[0][0]
            wc, ia = v["SQ_WAVE_CYCLES"] or 1, v["SQ_ACTIVE_INST_ANY"] or 1
            fo.write(f"{k:42s} {n:8d} {v['SQ_WAVES']:7.0f} {v['SQ_ACTIVE_INST_VALU'] / ia:9.2f} {v['SQ_WAIT_ANY'] / wc:9.2f} "
                     f"{v['SQ_WAIT_INST_ANY'] / wc:10.2f} {ia / wc:9.2f} {v['SQ_ACTIVE_INST_LDS'] / ia:8.2f}\n")


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
    for name in ("stats256", "stats512", "statsslab", "statsslab_p2p", "stats128", "stats64"):
        rows = kernel_stats(os.path.join(src, name))
        if not rows:
            continue
        with open(os.path.join(dst, f"{rnd}_{name}_kernel_stats.csv"), "w") as fo:
            fo.write("kernel,calls,avg_us,total_ms,percent,min_us,max_us\n")
            for r in rows:
                fo.write(f"\"{r[0]}\",{r[1]},{r[2]:.2f},{r[3]:.3f},{r[4]:.2f},{r[5]:.2f},{r[6]:.2f}\n")
        b = bench_line(os.path.join(src, name + ".log"))
        if b:
            json.dump(b, open(os.path.join(dst, f"{rnd}_{name}_bench.json"), "w"), indent=1)
        out[name] = rows
    sq_counters(os.path.join(src, "pmc256_SQ"), os.path.join(dst, f"{rnd}_sq_counters.txt"))
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
    # true read bytes: 128-B read requests are tallied at 64 B (x2).  (Until the reduced operators moved to the tile-blocked
    # layout their reads were 64-B requests, tallied exactly, and were excluded from the doubling.)
    op_bytes = 4 * P * n * n  # one imported reduced real operator
    op_reads = {}
    steps = None
    table = []
    for k in f256:
        if k not in w256:
            continue
        nops = op_reads.get(k, 0) * op_bytes
        rd = 2.0 * (f256[k][1] - nops) + nops
        table.append({"kernel": k, "launches": f256[k][0], "FETCH_SIZE_bytes": round(f256[k][1]),
                      "WRITE_SIZE_bytes": round(w256[k][1]), "read_bytes_corrected": round(rd),
                      "traffic_bytes": round(rd + w256[k][1])})
        if k.startswith("k_xinv<256, 3, true"):
            steps = f256[k][0]
    by = {t["kernel"]: t for t in table}
    for t in table:  # template arguments added over time (k_xinv<L, EPI, CHAIN, TERMS, TAIL>, k_xfwd<L, TAIL>): the keys
        # used below are the leading arguments; the unmasked (TAIL = false) instantiation is the one that matters
        if t["kernel"].endswith(", true>") and re.match(r"k_x(inv|fwd|shift)<", t["kernel"]) and t["kernel"].count(",") in (1, 4):
            continue
        m = re.match(r"(k_xinv<256, \d, (?:true|false))", t["kernel"])
        if m:
            by.setdefault(m.group(1) + ">", t)
        m = re.match(r"(k_xfwd<256)", t["kernel"])
        if m:
            by.setdefault("k_xfwd<256>", t)
    # the loop's kernels only: not the set-up (operator import, p0 source, initial velocity, first x-forward) and not
    # bench.py's copy-bandwidth probe
    per_step = sum(t["traffic_bytes"] * t["launches"] for t in table if t["kernel"].startswith("k_") and
                   not t["kernel"].startswith(("k_import", "k_add_initial", "k_xinv<256, 2", "k_stream_copy", "k_probe",
                                               "k_xinv<256, 4, false", "k_xfwd"))) / max(steps or 1, 1)

    def per_array(kernel, arrays_per_step):
        t = by.get(kernel)
        return t["traffic_bytes"] * t["launches"] / steps / arrays_per_step if t else 0.0
    yf, yi = per_array("k_ypass<256, -1, false, false>", 6), per_array("k_ypass<256, 1, false, false>", 7.5)
    g = lambda k: by.get(k, {}).get("traffic_bytes", 0)
    entry = {
        "fused_velocity": g("k_xfwd<256>") + yf + g("k_zfused<256, 0>") + 3 * yi + 3 * g("k_xinv<256, 1, true>") / 1 + 3 * yf,
        "fused_density": g("k_zfused<256, 1>") + 3 * yi + g("k_xinv<256, 3, true>") + 2 * yf,
        "fused_absorption_pressure": g("k_zfused<256, 2>") + 2 * yi + g("k_xinv<256, 4, false>") + g("k_xinv<256, 4, true>") + yf,
    }
    # k_xinv<256,1,true> is launched once per step with grid.y = 3: its per-launch figure already covers 3 components
    entry["fused_velocity"] = g("k_zfused<256, 0>") + 3 * yi + g("k_xinv<256, 1, true>") + 3 * yf
    bench_names = {"k_xfwd[1]": g("k_xfwd<256>"), "k_zfused_pgrad": g("k_zfused<256, 0>"),
                   "k_zfused_vgrad[3]": g("k_zfused<256, 1>"), "k_zfused_absorb[2]": g("k_zfused<256, 2>"),
                   "k_xinv_velocity_chain": g("k_xinv<256, 1, true>"), "k_xinv_density_chain": g("k_xinv<256, 3, true>"),
                   "k_xinv_psum": g("k_xinv<256, 4, false>"), "k_xinv_psum_chain": g("k_xinv<256, 4, true>")}
    bench_names["k_ypass_inv_pgrad[3]"] = 2.5 * yi
    for na in (1, 2, 3):
        bench_names[f"k_ypass_fwd[{na}]"] = na * yf
        bench_names[f"k_ypass_inv[{na}]"] = na * yi
    hash_file = os.path.join(src, "source_hash.txt")
    res = {"grid": [n, n, n], "units": "bytes per kernel launch; traffic = corrected reads + WRITE_SIZE",
           "kernel_source_hash": open(hash_file).read().strip() if os.path.exists(hash_file) else None,
           "calibration": calib, "fetch_ratio_8B_per_lane": r8, "fetch_ratio_16B_per_lane": r16,
           "per_kernel_launch": table, "steps_profiled": steps, "traffic_bytes_per_step": round(per_step),
           "traffic_bytes_per_entry_point": {k: round(v) for k, v in entry.items()},
           "traffic_bytes_per_bench_kernel": {k: round(v) for k, v in bench_names.items()}}
    json.dump(res, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("traffic_bytes_per_step", "traffic_bytes_per_entry_point", "steps_profiled")}, indent=1))
    for t in table:
        print(f"{t['kernel']:42s} n={t['launches']:3d} traffic={t['traffic_bytes'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
