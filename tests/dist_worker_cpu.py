"""Worker for tests/test_dist_cpu.py (world_size >= 2, gloo, CPU only).

Checks, per rank, (1) the slab partition of a problem and (2) a NumPy model of the slab-decomposed 3-D FFT that uses
exactly the packed-chunk addressing of csrc/kw_fused.hip (RowAddr "packed": row(z, ky) = ((ky / nyl) * nzl + z) * nyl +
ky % nyl) with torch.distributed.all_to_all_single as the transpose, against numpy.fft.rfftn of the whole grid.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.dist import partition_problem, slab_range  # noqa: E402


def a2a(send: np.ndarray) -> np.ndarray:
    t_in = torch.from_numpy(np.ascontiguousarray(send).view(np.float64).copy())
    t_out = torch.empty_like(t_in)
    dist.all_to_all_single(t_out, t_in)
    return t_out.numpy().view(np.complex128).reshape(send.shape)


def main():
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    nx, ny, nz = 16, 8 * P, 4 * P
    rng = np.random.default_rng(7)
    g = rng.standard_normal((nz, ny, nx))
    z0, z1 = slab_range(nz, rank, P)
    nzl, nyl, nxc = nz // P, ny // P, nx // 2 + 1
    # ---- forward: local x,y passes, pack per peer, all-to-all, z pass ----
    s = np.fft.fft(np.fft.rfft(g[z0:z1], axis=2), axis=1)                   # [nzl][ny][nxc]
    packed = np.empty((P, nzl, nyl, nxc), dtype=np.complex128)               # chunk q -> rank q
    for ky in range(ny):
        packed[ky // nyl, :, ky % nyl, :] = s[:, ky, :]
    recv = a2a(packed).reshape(nz, nyl, nxc)                                  # [src rank][zl] == global z
    spec = np.fft.fft(recv, axis=0)                                           # transposed spectrum [nz][nyl][nxc]
    ref = np.fft.rfftn(g, axes=(0, 1, 2))[:, rank * nyl:(rank + 1) * nyl, :]
    err_f = np.abs(spec - ref).max() / np.abs(ref).max()
    # ---- inverse: z pass, all-to-all back (chunks are contiguous z ranges), unpack, y and x passes ----
    back = a2a(np.fft.ifft(spec, axis=0).reshape(P, nzl, nyl, nxc))           # [src q][zl][kyl][nxc]
    t = np.empty((nzl, ny, nxc), dtype=np.complex128)
    for ky in range(ny):
        t[:, ky, :] = back[ky // nyl, :, ky % nyl, :]
    loc = np.fft.irfft(np.fft.ifft(t, axis=1), n=nx, axis=2)
    err_i = np.abs(loc - g[z0:z1]).max()
    # ---- partition of a problem: local pieces reassemble to the global problem ----
    pr = synthetic.make_problem(16, 8 * P, 4 * P, source="p_source", source_many=1, nt=6, pml_size=2, sensor="random")
    loc_pr, info = partition_problem(pr, rank, P)
    assert int(loc_pr["Nz"].ravel()[0]) == nzl
    assert loc_pr["c0"].shape == (nzl, 8 * P, 16) and np.array_equal(loc_pr["c0"], pr["c0"][z0:z1])
    assert np.array_equal(loc_pr["pml_z"], pr["pml_z"][z0:z1])
    assert np.array_equal(loc_pr["ddz_k_shift_pos"], pr["ddz_k_shift_pos"])  # 1-D k-space operators stay global
    gi = pr["p_source_index"].reshape(-1).astype(np.int64) - 1
    li = loc_pr["p_source_index"].reshape(-1).astype(np.int64) - 1
    sel = np.nonzero((gi >= z0 * 16 * 8 * P) & (gi < z1 * 16 * 8 * P))[0]
    assert np.array_equal(li + z0 * 16 * 8 * P, gi[sel])
    assert np.array_equal(loc_pr["p_source_input"].reshape(6, -1), pr["p_source_input"].reshape(6, -1)[:, sel])
    counts = torch.tensor([li.size, info["sensor_positions"].size], dtype=torch.int64)
    dist.all_reduce(counts)
    assert int(counts[0]) == gi.size and int(counts[1]) == pr["sensor_mask_index"].size
    slab_pr = synthetic.make_problem(16, 8 * P, 4 * P, source="p_source", source_many=1, nt=6, pml_size=2,
                                     sensor="random", zslab=(z0, z1))
    loc2, _ = partition_problem(slab_pr, rank, P, arrays_are_local=True)
    for k in loc_pr:
        assert np.array_equal(loc_pr[k], loc2[k]), k
    ok = err_f < 1e-12 and err_i < 1e-12
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"DIST_CPU_OK={int(flag)} err_fwd={err_f:.2e} err_inv={err_i:.2e}")
    dist.destroy_process_group()
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
