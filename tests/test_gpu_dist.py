"""Multi-rank slab-decomposed runs on the one-GPU box (P processes share the card, gloo all-to-all through the host):
the N>1 code path — partition, packed y-pass, transposed z-pass, exchange callback — against the CPU oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-5


def run_ranks(world, dims, steps, source, mode, tmp_path, backend="gloo", medium="111"):
    out = str(tmp_path / f"dist_{world}_{source}_{mode}_{backend}_{medium}.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29700 + world + 10 * mode),
           os.path.join(HERE, "dist_worker_gpu.py"), "--dims", *map(str, dims), "--steps", str(steps), "--source", source,
           "--mode", str(mode), "--backend", backend, "--medium", medium, "--out", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-4000:]
    return np.load(out)


@pytest.mark.parametrize("world,dims,source,mode", [
    (2, (32, 32, 32), "p0", 0),
    (4, (32, 64, 64), "p0", 0),
    (2, (32, 32, 32), "p_source", 2),   # k-space corrected additive source: scaleSource is collective
    (2, (64, 32, 16), "u_source", 1),
    (2, (16, 512, 16), "p0", 0),        # 512-point y lines (2 x 256 kernels) with packed per-peer addressing
    (2, (16, 16, 512), "p0", 0),        # 512-point z lines on the transposed spectra
    (2, (48, 96, 80), "p0", 0),         # radix-3 / radix-5 lines; 48 ky rows per rank (no power of two)
    (4, (32, 120, 48), "p_source", 2),  # 30 ky rows, 12 planes per rank
])
def test_slab_ranks_match_oracle(orc, syn, tmp_path, world, dims, source, mode):
    steps = 20
    res = run_ranks(world, dims, steps, source, mode, tmp_path)
    nx, ny, nz = dims
    pr = syn.make_problem(nx, ny, nz, heterogeneous=True, nonlinear=True, absorbing=True, source=source,
                          source_mode=mode, source_many=1, nt=steps, pml_size=4, sensor="random")
    o = orc.OracleSim(pr)
    series = []
    for _ in range(steps):
        o.step()
        series.append(o.field("p").reshape(-1)[o.sensor_index].copy())
    for f in ("p", "ux", "uz", "rhoy"):
        assert rel_l2(res[f], o.field(f)) < TOL, f
    assert rel_l2(res["series"], np.array(series)) < TOL
    assert int(res["exchanges"][0]) > 10 * steps  # the all-to-all really ran (14 per absorbing step)
    o.close()


@pytest.mark.parametrize("dims,source,mode", [((32, 32, 32), "p0", 0), ((64, 32, 16), "p_source", 2)])
def test_slab_path_over_rccl_single_rank(orc, syn, tmp_path, dims, source, mode):
    """backend nccl (= RCCL) with one rank: the slab code path exchanging with itself — device-resident scratch tensors,
    all_to_all_single(async_op=True) / work.wait() on the solver's stream, split-phase pipelining — on real RCCL."""
    steps = 20
    res = run_ranks(1, dims, steps, source, mode, tmp_path, backend="nccl")
    nx, ny, nz = dims
    pr = syn.make_problem(nx, ny, nz, heterogeneous=True, nonlinear=True, absorbing=True, source=source,
                          source_mode=mode, source_many=1, nt=steps, pml_size=4, sensor="random")
    o = orc.OracleSim(pr)
    series = []
    for _ in range(steps):
        o.step()
        series.append(o.field("p").reshape(-1)[o.sensor_index].copy())
    for f in ("p", "ux", "uz", "rhoy"):
        assert rel_l2(res[f], o.field(f)) < TOL, f
    assert rel_l2(res["series"], np.array(series)) < TOL
    assert int(res["exchanges"][0]) > 10 * steps
    o.close()


@pytest.mark.parametrize("world,dims,source,mode,medium", [
    (2, (32, 32, 32), "p0", 0, "100"),        # linear lossless: the equation of state runs inside the density kernel
    (2, (32, 64, 32), "p_source", 1, "010"),  # homogeneous nonlinear lossless with an additive source
    (4, (32, 32, 64), "u_source", 2, "001"),  # homogeneous linear absorbing, k-space corrected velocity source
])
def test_slab_ranks_other_media(orc, syn, tmp_path, world, dims, source, mode, medium):
    steps = 16
    res = run_ranks(world, dims, steps, source, mode, tmp_path, medium=medium)
    nx, ny, nz = dims
    het, nonlin, absorb = (c == "1" for c in medium)
    pr = syn.make_problem(nx, ny, nz, heterogeneous=het, nonlinear=nonlin, absorbing=absorb, source=source,
                          source_mode=mode, source_many=1, nt=steps, pml_size=4, sensor="random")
    o = orc.OracleSim(pr)
    o.step(steps)
    for f in ("p", "ux", "uz", "rhoy"):
        assert rel_l2(res[f], o.field(f)) < TOL, f
    o.close()
