#!/usr/bin/env python3
"""per-kernel SQ / LDS counter table of gpurun_out/r03_pmclen_<n>_<pass>/ (tools/r03_pmc_len.sh): python tools/pmc_len_table.py 240 256"""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLOCK_GHZ = 2.4  # waves/CU = 4 * SQ_WAVE_CYCLES (counted in quad-cycles) / (launch duration * clock * 256 CUs): resident waves per CU, approximately
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", name).strip()
for n in sys.argv[1:]:
    acc = collections.OrderedDict()
    dur = collections.defaultdict(lambda: [0, 0.0])  # kernel -> launches, total ns (under the counter pass: serialised launches)
    for d in sorted(glob.glob(f"{ROOT}/gpurun_out/r03_pmclen_{n}_1/")):
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                dur[k][0] += 1
                dur[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for d in sorted(glob.glob(f"{ROOT}/gpurun_out/r03_pmclen_{n}_*/")):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc.setdefault(short(r["Kernel_Name"]), collections.OrderedDict())
                c = a.setdefault(r["Counter_Name"], [0, 0.0])
                c[0] += 1
                c[1] += float(r["Counter_Value"])
    print(f"== {n}^3  (per launch; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES, ldsw = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, conf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE,")
    print("    lds/busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CYCLES, valu/wave, vmem/wave: active-instruction cycles per wave cycle)")
    print(f"{'kernel':44s} {'waves':>7s} {'wavecyc/wave':>12s} {'wait':>5s} {'ldsw':>5s} {'conf':>5s} {'lds/busy':>8s} {'valu/wave':>9s} {'vmem/wave':>9s} {'VALU insts/wave':>15s} {'us':>7s} {'waves/CU':>8s}")
    for k, v in acc.items():
        if not k.startswith("k_") or "stream_copy" in k or "import" in k or "probe" in k:
            continue
        g = lambda c: (v[c][1] / v[c][0]) if c in v else float("nan")
        wc = g("SQ_WAVE_CYCLES") or 1
        print(f"{k[:44]:44s} {g('SQ_WAVES'):7.0f} {wc / max(g('SQ_WAVES'), 1):12.0f} {g('SQ_WAIT_ANY') / wc:5.2f} {g('SQ_WAIT_INST_LDS') / wc:5.2f} "
              f"{g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):5.2f} {g('SQ_LDS_IDX_ACTIVE') / max(g('SQ_BUSY_CYCLES'), 1):8.3f} "
              f"{g('SQ_ACTIVE_INST_VALU') / wc:9.3f} {g('SQ_ACTIVE_INST_VMEM') / wc:9.3f} {g('SQ_INSTS_VALU') / max(g('SQ_WAVES'), 1):15.0f} "
              f"{(dur[k][1] / max(dur[k][0], 1)) / 1e3:7.1f} {4.0 * wc / max((dur[k][1] / max(dur[k][0], 1)) * CLOCK_GHZ * 256, 1):8.1f}")
