#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD container, where /root/reference exists).

  compress_ref.npz   outputs of the REFERENCE's own Compression/CompressHelper.cpp, compiled from where it lies by
                     oracle/Makefile into oracle/_ref/libcompress_ref.so (never copied): the four bases for three
                     (period, mos, harmonics) triples, findPeriod on sampled sines, 40-bit codec round trips.
                     These pin oracle/kwave_oracle.c's restatement of the compression basis (config 5).
  oracle_cases.npz   outputs of the CPU oracle (sensor series + final-field checksums) for small seeded problems:
                     regression pins for the oracle itself and reference data for the GPU path on the GPU box, where
                     neither /root/reference nor a long oracle run is available.

Fixtures are data only (inputs are regenerated from kwave_amd.synthetic with the recorded arguments).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = {
    "c32_het_nl_abs_p0": dict(args=(32,), kw=dict(heterogeneous=True, nonlinear=True, absorbing=True, source="p0",
                                                  pml_size=6, sensor="random"), steps=40),
    "c32_hom_lin_lossless_p0": dict(args=(32,), kw=dict(heterogeneous=False, nonlinear=False, absorbing=False,
                                                         source="p0", pml_size=6, sensor="random"), steps=40),
    "c32_het_lin_abs_psrc_additive": dict(args=(32,), kw=dict(heterogeneous=True, nonlinear=False, absorbing=True,
                                                               source="p_source", source_mode=2, source_many=1, nt=40,
                                                               pml_size=4, sensor="random"), steps=40),
    "c24x20x18_het_nl_abs_usrc": dict(args=(24, 20, 18), kw=dict(heterogeneous=True, nonlinear=True, absorbing=True,
                                                                  source="u_source", source_mode=1, source_many=0, nt=30,
                                                                  pml_size=4, sensor="random"), steps=30),
}
COMPRESS = [(10.0, 1, 1), (12.5, 1, 2), (20.0, 2, 3)]


def make_compress():
    orc.build(force=False)
    if not os.path.exists(orc.REF_LIB_PATH):
        raise SystemExit("oracle/_ref/libcompress_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    L = C.CDLL(orc.REF_LIB_PATH)
    L.cref_init.argtypes = [C.c_float, C.c_ulonglong, C.c_ulonglong]
    L.cref_osize.restype = C.c_ulonglong
    L.cref_bsize.restype = C.c_ulonglong
    L.cref_basis.argtypes = [C.c_int, C.c_void_p]
    L.cref_find_period.restype = C.c_float
    L.cref_find_period.argtypes = [C.c_void_p, C.c_ulonglong]
    L.cref_to40b.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int]
    L.cref_from40b.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    out = {}
    for i, (period, mos, harm) in enumerate(COMPRESS):
        L.cref_init(period, mos, harm)
        bs, osz = int(L.cref_bsize()), int(L.cref_osize())
        out[f"cfg{i}_params"] = np.array([period, mos, harm, osz, bs], dtype=np.float64)
        for which, nm in enumerate(("bE", "bE_1", "bEShifted", "bE_1Shifted")):
            buf = np.empty((harm, bs, 2), dtype=np.float32)
            L.cref_basis(which, buf.ctypes.data)
            out[f"cfg{i}_{nm}"] = buf
    # findPeriod on sampled sines (Parameters.cpp:495-508 feeds it the last <= 500 samples of p_source_input)
    periods_in = np.array([8.0, 12.5, 20.0, 33.3], dtype=np.float64)
    found = []
    for per in periods_in:
        x = np.sin(2 * np.pi * np.arange(400) / per).astype(np.float32)
        found.append(L.cref_find_period(x.ctypes.data, x.size))
    out["find_period_in"] = periods_in
    out["find_period_out"] = np.array(found, dtype=np.float32)
    # 40-bit codec (CompressHelper.cpp:224-389), exponents kMaxExpP = 138, kMaxExpU = 114
    rng = np.random.default_rng(40)
    for e in (138, 114):
        scale = 1.0e6 if e == 138 else 1.0
        vals = (rng.standard_normal((256, 2)) * scale * 10.0 ** rng.uniform(-6, 0, size=(256, 1))).astype(np.float32)
        packed = np.zeros((256, 5), dtype=np.uint8)
        back = np.zeros((256, 2), dtype=np.float32)
        for j in range(256):
            L.cref_to40b(float(vals[j, 0]), float(vals[j, 1]), packed[j].ctypes.data, e)
            L.cref_from40b(packed[j].ctypes.data, back[j].ctypes.data, e)
        out[f"codec{e}_in"], out[f"codec{e}_packed"], out[f"codec{e}_out"] = vals, packed, back
    np.savez_compressed(os.path.join(HERE, "compress_ref.npz"), **out)
    print("wrote compress_ref.npz", {k: v.shape for k, v in out.items() if k.startswith("cfg0")})


def make_oracle_cases():
    out = {}
    for name, case in CASES.items():
        pr = synthetic.make_problem(*case["args"], **case["kw"])
        o = orc.OracleSim(pr)
        series = []
        for _ in range(case["steps"]):
            o.step()
            series.append(o.field("p").reshape(-1)[o.sensor_index].copy())
        out[name + "_series"] = np.array(series, dtype=np.float32)[:, ::8]  # every 8th sensor point keeps the file small
        for f in ("p", "ux", "rhoz"):
            a = o.field(f).astype(np.float64)
            out[name + "_" + f + "_norm"] = np.array([np.linalg.norm(a), a.sum(), np.abs(a).max()])
        # a coarse sub-sampled copy of the final pressure (every 4th point) for a direct field comparison
        out[name + "_p_sub"] = o.field("p")[::4, ::4, ::4].copy()
        o.close()
    np.savez_compressed(os.path.join(HERE, "oracle_cases.npz"), **out)
    print("wrote oracle_cases.npz", sum(v.nbytes for v in out.values()), "bytes uncompressed")


if __name__ == "__main__":
    make_compress()
    make_oracle_cases()
