"""Consistency of the fast path's line-length tables (CPU only: parses the sources): every length of KW_FUSED_LENGTHS has a
factor pair with both factors register-DFT sizes, the tuning tables name only listed lengths, and README / DESIGN quote the
same list."""
import os
import re

from conftest import ROOT

CSRC = os.path.join(ROOT, "k-wave-fluid-cuda_amd", "csrc")


def _lengths():
    src = open(os.path.join(CSRC, "kw_fused.hip")).read()
    short = re.search(r"#define KW_FUSED_LENGTHS_SHORT\(X\)(.*?)\n#define", src, re.S).group(1)
    long_ = re.search(r"#define KW_FUSED_LENGTHS_LONG\(X\)(.*?)\n#if", src, re.S).group(1)
    return [int(x) for x in re.findall(r"X\((\d+)\)", short)], [int(x) for x in re.findall(r"X\((\d+)\)", long_)], src


def _pairs():
    hdr = open(os.path.join(CSRC, "kw_fft_device.h")).read()
    pairs = {int(m.group(1)): (int(m.group(2)), int(m.group(3)))
             for m in re.finditer(r"struct Fac0<(\d+)>\s*\{ static constexpr int R1 = (\d+),\s*R2 = (\d+)", hdr)}
    swapped = {int(x) for x in re.findall(r"case (\d+):", re.search(r"constexpr bool fac_swapped.*?\n}\n", hdr, re.S).group(0))}
    return pairs, swapped


def test_every_listed_length_has_a_register_factor_pair():
    short, long_, src = _lengths()
    pairs, swapped = _pairs()
    split = int(re.search(r"#define KW_LONG_LINES (\d+)", src).group(1))
    assert short == sorted(short) and long_ == sorted(long_) and max(short) < split <= min(long_)
    dft_sizes = {2 ** m * odd for m in range(0, 6) for odd in (1, 3, 5, 7, 9, 15, 25, 27) if 2 ** m * odd <= 32}
    for n in short + long_:
        assert n in pairs, n
        r1, r2 = pairs[n]
        assert r1 * r2 == n and r1 in dft_sizes and r2 in dft_sizes, (n, r1, r2)
        assert n % 4 == 0  # the x kernels move float4
    assert set(pairs) == set(short + long_)
    assert swapped <= set(short + long_)
    for n in swapped:
        assert pairs[n][0] != pairs[n][1], n  # a square pair has no orientation


def test_tuning_tables_and_documents_name_listed_lengths_only():
    short, long_, src = _lengths()
    listed = set(short + long_)
    nlx = re.search(r"constexpr int nl_x\(int L\)\n\{.*?\n\}\n", src, re.S).group(0)
    assert {int(x) for x in re.findall(r"case (\d+):", nlx)} <= listed
    no_tail = re.search(r"constexpr bool has_partial_x_tiles\(int L\)\n\{.*?\n\}\n", src, re.S).group(0)
    assert {int(x) for x in re.findall(r"case (\d+):", no_tail)} <= set(long_)
    readme = open(os.path.join(ROOT, "README.md")).read()
    block = re.search(r"sides are each one of ([\d\s]+)\(2-D", readme).group(1)
    assert [int(x) for x in block.split()] == sorted(listed)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    block = re.search(r"each of Nx, Ny, Nz ∈ \{([\d,\s]+)\}", design).group(1)
    assert [int(x) for x in block.replace(",", " ").split()] == sorted(listed)
