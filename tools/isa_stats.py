#!/usr/bin/env python3
"""Per-kernel instruction mix of kw_fused.hip (gfx950), from hipcc -S.  Offline: no GPU needed.

  python tools/isa_stats.py [extra hipcc flags...]     e.g.  -DKW_PK=1
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "k-wave-fluid-cuda_amd", "csrc", "kw_fused.hip")


def main():
    extra = sys.argv[1:]
    out = os.environ.get("KW_ISA_OUT", "/tmp/kw_fused_isa.s")
    # the file is built in five passes (KW_FUSED_TU = 0 ... 4: see its header); all are listed
    procs, outs = [], []
    for tu in (0, 1, 2, 3, 4):
        o = f"{out}.tu{tu}"
        outs.append(o)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize",
                                       "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
                                       "-I" + os.path.dirname(SRC), SRC, "-o", o, f"-DKW_FUSED_TU={tu}"] + extra))
    for pr in procs:
        if pr.wait() != 0:
            raise SystemExit("hipcc failed")
    with open(out, "w") as f:
        for o in outs:
            f.write(open(o).read())
    kern, stats, meta = None, collections.OrderedDict(), {}
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
            stats[kern] = collections.Counter()
            continue
        if kern is None:
            continue
        s = line.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            pass
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if m and not line.strip().startswith("."):
            op = m.group(1)
            c = stats[kern]
            c["total"] += 1
            if op.startswith("v_pk_"): c["v_pk"] += 1
            if op.startswith("v_mov") or op.startswith("v_accvgpr"): c["v_mov"] += 1
            if op.startswith("v_"): c["valu"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
                c["vmem"] += 1
                if op.startswith("scratch_"): c["scratch"] += 1
            if op in ("s_waitcnt",): c["waitcnt"] += 1
            if op == "s_barrier": c["barrier"] += 1
        m = re.match(r"^\s+(?:- )?\.(vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|name):\s+(\S+)", line)
        if m:
            cur = meta.setdefault("pending", {})
            cur[m.group(1)] = m.group(2)
        if re.match(r"^\s+\.wavefront_size:", line) and "pending" in meta:
            cur = meta.pop("pending")
            meta[cur.get("name")] = {k: int(v) for k, v in cur.items() if k != "name"}
    dem = subprocess.run(["c++filt"] + list(stats), capture_output=True, text=True).stdout.split("\n")
    print(f"{'kernel':46s} {'total':>6s} {'valu':>6s} {'v_pk':>5s} {'v_mov':>5s} {'salu':>5s} {'lds':>5s} {'vmem':>5s} {'scr':>4s} {'bar':>4s} | vgpr agpr spill lds")
    for (k, c), d in zip(stats.items(), dem):
        if not c["total"]:
            continue
        d = d.replace("(anonymous namespace)::", "").replace("void ", "")
        d = re.sub(r"\(.*\)$", "", d)
        mm = meta.get(k, {})
        print(f"{d[:46]:46s} {c['total']:6d} {c['valu']:6d} {c['v_pk']:5d} {c['v_mov']:5d} {c['salu']:5d} {c['lds']:5d} {c['vmem']:5d} {c['scratch']:4d} {c['barrier']:4d} |"
              f" {mm.get('vgpr_count', -1):4d} {mm.get('agpr_count', -1):4d} {mm.get('vgpr_spill_count', -1):5d} {mm.get('group_segment_fixed_size', -1)}")


if __name__ == "__main__":
    main()
