#!/usr/bin/env python3
"""Per-length tuning table from a tools/abk.sh log of the variant matrix (builds L<len>_<b|s>_n<NLX>: factor orientation as
listed / swapped, x-tile height NLX line pairs): python tools/length_tuning_table.py gpurun_out/r03_abk5.log [--emit]
Prints steps/s of every variant per length, the best one, the gain over the round-2 choice (b, n16 below 400 / n8 from 400),
and with --emit the C++ switch bodies for fac_swapped() / nl_x()."""
import collections, re, sys
rows = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r"L(\d+)_([bs])_n(\d+)(\S*) (\d) ([\d.]+) (\{.*\})", l)
    if not m:
        continue
    L, o, nl, extra, val = int(m.group(1)), m.group(2), int(m.group(3)), m.group(4), float(m.group(6))
    rows.setdefault(L, collections.OrderedDict()).setdefault((o, nl, extra), []).append(val)
best = {}
print(f"{'L':>5s}  " + "  ".join(f"{o}_n{nl:<2d}" for o in "bs" for nl in (8, 10, 12, 16)) + "   best   gain over the round-2 form")
for L, v in rows.items():
    med = {k: sorted(x)[len(x) // 2] for k, x in v.items()}
    old = med.get(("b", 8 if L >= 400 else 16, ""))
    k = max((k for k in med if k[2] == ""), key=lambda k: med[k])
    best[L] = k
    cells = []
    for o in "bs":
        for nl in (8, 10, 12, 16):
            cells.append(f"{med[(o, nl, '')]:7.1f}" if (o, nl, "") in med else "      -")
    gain = f"{(med[k] / old - 1) * 100:+5.1f} %" if old else "   ?"
    print(f"{L:5d}  " + " ".join(cells) + f"   {k[0]}_n{k[1]:<2d}  {gain}")
    for kk, x in med.items():
        if kk[2]:
            print(f"       {kk[0]}_n{kk[1]}{kk[2]}: {x:.1f}")
if "--emit" in sys.argv:
    sw = sorted(L for L, k in best.items() if k[0] == "s")
    print("\n// fac_swapped: case " + ": case ".join(str(L) for L in sw) + ": return true;")
    by = collections.defaultdict(list)
    for L, k in sorted(best.items()):
        by[k[1]].append(L)
    for nl, Ls in sorted(by.items()):
        print(f"// nl_x: case " + ": case ".join(str(L) for L in Ls) + f": return {nl};")
